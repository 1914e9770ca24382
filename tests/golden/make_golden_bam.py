"""Golden BAM files from the REAL reference front end (`oracle/_ref/ngskit4b kalign -o <case>.bam`, built by `make -C oracle ngskit4b`).

    python tests/golden/make_golden_bam.py

For cases of make_golden_sam.py (same reads: tests/golden/sam_<case>*.fa.xz, same arguments: sam_cases.json) the BAM and the
.bai the reference wrote.  Checked here before they are kept: the BAM, decoded (tests/samutil.py), holds exactly the records of
the SAM golden of the same case.  Data only."""
import json
import lzma
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(HERE))
import samutil  # noqa: E402

NGS = os.path.join(ROOT, "oracle", "_ref", "ngskit4b")
CASES = ["se_s2", "pe_u1", "se_all_120"]
ALL_READS_CASES = ["se_s2_M1", "pe_u1_M1"]  # -M1 (sam_all_cases.json): the unaligned records follow the alignments


def main():
    meta = json.load(open(os.path.join(HERE, "sam_cases.json")))
    with tempfile.TemporaryDirectory() as tmp:
        for case in CASES:
            m = meta[case]
            index = m.get("index", "g1")
            sfx = os.path.join(tmp, index + ".sfx")
            if not os.path.exists(sfx):
                if os.path.exists(os.path.join(HERE, index + ".sfx")):
                    shutil.copy(os.path.join(HERE, index + ".sfx"), sfx)
                else:
                    with lzma.open(os.path.join(HERE, index + ".sfx.xz")) as f, open(sfx, "wb") as g:
                        g.write(f.read())
            files = []
            if case.startswith("pe_"):
                for k, flag in (("1", "-i"), ("2", "-u")):
                    fa = os.path.join(tmp, "%s_%s.fa" % (case, k))
                    with lzma.open(os.path.join(HERE, "sam_%s_%s.fa.xz" % (case, k))) as f, open(fa, "wb") as g:
                        g.write(f.read())
                    files += [flag, fa]
            else:
                fa = os.path.join(tmp, case + ".fa")
                with lzma.open(os.path.join(HERE, m.get("reads", "sam_%s.fa.xz" % case))) as f, open(fa, "wb") as g:
                    g.write(f.read())
                files = ["-i", fa]
            bam = os.path.join(tmp, case + ".bam")
            subprocess.run([NGS, "kalign", "-I", sfx, "-o", bam, "-T", "4", "-F", os.path.join(tmp, case + ".log")] + m["args"] + files,
                           check=True, capture_output=True)
            hdr, refs, recs = samutil.read_bam(bam)
            _, sam_recs = samutil.read_sam_xz(os.path.join(HERE, "sam_%s.sam.xz" % case))
            as_sam = [samutil.bam_record_as_sam_fields(r, refs) for r in recs]
            assert sorted(as_sam) == sorted(samutil.sam_line_fields(x) for x in sam_recs), case
            shutil.copy(bam, os.path.join(HERE, "bam_%s.bam" % case))
            shutil.copy(bam + ".bai", os.path.join(HERE, "bam_%s.bam.bai" % case))
            print(case, len(recs), "records,", os.path.getsize(bam), "bytes")
        all_meta = json.load(open(os.path.join(HERE, "sam_all_cases.json")))
        for case in ALL_READS_CASES:
            m = all_meta[case]
            base = m["reads_of"]
            files = []
            for flag, suffix in (("-i", "_1"), ("-u", "_2")) if base.startswith("pe_") else (("-i", ""),):
                fa = os.path.join(tmp, "%s%s.all.fa" % (base, suffix))
                with lzma.open(os.path.join(HERE, "sam_%s%s.fa.xz" % (base, suffix))) as f, open(fa, "wb") as g:
                    g.write(f.read())
                files += [flag, fa]
            bam = os.path.join(tmp, case + ".bam")
            subprocess.run([NGS, "kalign", "-I", os.path.join(HERE, "g1.sfx"), "-o", bam, "-T", "4", "-F", os.path.join(tmp, case + ".log")] + m["args"] + files,
                           check=True, capture_output=True)
            hdr, refs, recs = samutil.read_bam(bam)
            _, sam_recs = samutil.read_sam_xz(os.path.join(HERE, "sam_%s.sam.xz" % case))
            assert len(recs) == len(sam_recs) and sum(1 for r in recs if r["ref"] < 0) == len(recs) - m["nar"]["AA"], case
            shutil.copy(bam, os.path.join(HERE, "bam_%s.bam" % case))
            shutil.copy(bam + ".bai", os.path.join(HERE, "bam_%s.bam.bai" % case))
            print(case, len(recs), "records,", os.path.getsize(bam), "bytes")


if __name__ == "__main__":
    main()
