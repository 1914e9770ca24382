"""CPU-side checks of the product boundary: the C-ABI library loads, exports every symbol include/k4sfx.h declares,
and refuses to compute without a GPU (no silent fallback)."""
import ctypes as C
import os
import re

import pytest

import kit4b_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    if not os.path.exists(kit4b_amd.LIB_PATH):
        kit4b_amd.build()
    return kit4b_amd.lib()


def test_header_symbols_all_exported(L):
    hdr = open(os.path.join(ROOT, "include", "k4sfx.h")).read()
    declared = set(re.findall(r"^(?:int|void|double|const char\*)\s+(k4_\w+)\s*\(", hdr, flags=re.M))
    assert declared == set(kit4b_amd.ABI_SYMBOLS), declared ^ set(kit4b_amd.ABI_SYMBOLS)
    for s in declared:
        assert hasattr(L, s), s
    assert L.k4_abi_version() == 2


def test_every_abi_symbol_has_its_argument_types_declared(L):
    """ctypes passes an undeclared Python int as a 32-bit C int: a device address would be truncated.  Every symbol of the
    ABI carries argtypes, and their number equals the parameter count of the header's declaration."""
    hdr = open(os.path.join(ROOT, "include", "k4sfx.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    for s in kit4b_amd.ABI_SYMBOLS:
        fn = getattr(L, s)
        assert fn.argtypes is not None, s
        m = re.search(r"\b%s\s*\(([^;]*?)\)\s*;" % s, hdr, flags=re.S)
        assert m, s
        params = m.group(1).strip()
        n = 0 if params in ("", "void") else params.count(",") + 1
        assert len(fn.argtypes) == n, (s, len(fn.argtypes), n)


def test_struct_sizes_match_header():
    assert C.sizeof(kit4b_amd.AlignParams) == 44  # ABI 2: + MinChimericLen, microInDelLen, MaxSpliceJunctLen
    assert C.sizeof(kit4b_amd.KalignParams) == 48
    assert kit4b_amd.SEG2_DTYPE.itemsize == 16
    assert kit4b_amd.HIT_DTYPE.itemsize == 16
    assert kit4b_amd.RESULT_DTYPE.itemsize == 24
    assert C.sizeof(kit4b_amd.Counters) == 48


def test_error_codes_follow_reference_values():
    hdr = open(os.path.join(ROOT, "include", "k4sfx.h")).read()
    vals = dict(re.findall(r"(K4_ERR_\w+)\s*=\s*(-?\d+)", hdr))
    # teBSFrsltCodes (libkit4b/ErrorCodes.h:15-97): eBSFerrParams=-100, Mem=-95, NotBioseq=-94, OpnFile=-90, ...
    assert vals["K4_ERR_PARAMS"] == "-100" and vals["K4_ERR_MEM"] == "-95" and vals["K4_ERR_NOT_SFX"] == "-94"
    assert vals["K4_ERR_OPEN_FILE"] == "-90" and vals["K4_ERR_FILE_VER"] == "-86" and vals["K4_ERR_ENTRY"] == "-51"


def test_open_errors_are_loud(L, golden_dir, tmp_path):
    import torch

    h = C.c_void_p()
    assert L.k4_open(b"/nonexistent/x.sfx", 0, 0, C.byref(h)) == -90  # eBSFerrOpnFile
    bad = tmp_path / "bad.sfx"
    bad.write_bytes(b"nope" * 400)
    assert L.k4_open(str(bad).encode(), 0, 0, C.byref(h)) == -94  # eBSFerrNotBioseq
    assert b"magic" in L.k4_global_error()
    if not torch.cuda.is_available():
        rc = L.k4_open(os.path.join(golden_dir, "g1.sfx").encode(), 0, 0, C.byref(h))
        assert rc == -2 and not h.value  # K4_ERR_NO_DEVICE: no CPU fallback
        assert b"no CPU fallback" in L.k4_global_error()


def test_product_does_not_reference_the_oracle():
    """The oracle is test infrastructure: nothing under kit4b_amd/ or include/ may mention it."""
    for base in ("kit4b_amd", "include"):
        for dp, _, fns in os.walk(os.path.join(ROOT, base)):
            for fn in fns:
                if fn.endswith((".so", ".o", ".pyc")):
                    continue
                txt = open(os.path.join(dp, fn), errors="ignore").read()
                assert "k4oracle" not in txt and "oracle_bindings" not in txt and "libk4ref" not in txt, (dp, fn)


def test_comm_library_exports_its_header():
    """include/k4comm.h (the RCCL side: index broadcast over xGMI, all-reduce of the tallies) <-> kit4b_amd/libk4comm.so"""
    hdr = open(os.path.join(ROOT, "include", "k4comm.h")).read()
    declared = set(re.findall(r"^(?:int|void|const char\*)\s+(k4_comm_\w+)\s*\(", hdr, flags=re.M))
    assert len(declared) == 10
    so = os.path.join(ROOT, "kit4b_amd", "libk4comm.so")
    assert os.path.exists(so), "make -C kit4b_amd/csrc"
    kit4b_amd.lib()  # libk4sfx.so first: libk4comm.so is linked against it
    Lc = C.CDLL(so)
    for s in declared:
        assert hasattr(Lc, s), s
    # RCCL is a dependency of this library only: the hot-path library stays free of it
    import subprocess

    needed = subprocess.run(["readelf", "-d", os.path.join(ROOT, "kit4b_amd", "libk4sfx.so")], capture_output=True, text=True).stdout
    assert "rccl" not in needed and "rccl" in subprocess.run(["readelf", "-d", so], capture_output=True, text=True).stdout


def test_index_exchange_schedule_for_every_rank_count():
    """k4_comm_bcast_schedule (include/k4comm.h): the list of transfers the RCCL calls of k4_comm_open_index are issued from.
    Simulated here for 1..8 ranks (and 64) and awkward sizes: a rank only sends what it holds when the phase starts, nothing is
    written twice, every rank ends with every byte, no (src, dst) link carries more than one piece per phase, the root's
    links carry exactly one piece each in phase 0 (all of its direct links busy, none twice)."""
    kit4b_amd.lib()
    Lc = C.CDLL(os.path.join(ROOT, "kit4b_amd", "libk4comm.so"))

    class Xfer(C.Structure):
        _fields_ = [("phase", C.c_int32), ("src", C.c_int32), ("dst", C.c_int32), ("off", C.c_uint64), ("len", C.c_uint64)]

    Lc.k4_comm_bcast_schedule.argtypes = [C.c_int, C.c_uint64, C.POINTER(Xfer), C.c_int]
    for n in (1, 2, 3, 4, 5, 6, 7, 8, 64):
        for size in (0, 1, 255, 256, 257, 4096 * n + 3, 1_000_003, 3_000_000_024, 15_000_000_120 * 5):
            cap = 2 * n * n
            buf = (Xfer * cap)()
            k = Lc.k4_comm_bcast_schedule(n, size, buf, cap)
            assert 0 <= k <= cap
            assert Lc.k4_comm_bcast_schedule(n, size, buf, 0) == k  # the count alone
            xs = [(x.phase, x.src, x.dst, x.off, x.len) for x in buf[:k]]
            if n == 1 or size == 0:
                assert k == 0
                continue
            piece = ((size + n - 1) // n + 255) & ~255
            have = [[(0, size)] if r == 0 else [] for r in range(n)]  # byte intervals each rank holds

            def holds(r, off, ln):
                return any(a <= off and off + ln <= b for a, b in have[r])

            for phase in (0, 1):
                cur = [x for x in xs if x[0] == phase]
                links = {}
                for _, src, dst, off, ln in cur:
                    assert src != dst and 0 <= src < n and 0 <= dst < n and ln > 0 and off + ln <= size
                    assert holds(src, off, ln), (n, size, phase, src, dst)          # held BEFORE the phase (sends of a phase run together)
                    assert not any(a < off + ln and off < b for a, b in have[dst])  # nothing arrives twice
                    links[(src, dst)] = links.get((src, dst), 0) + ln
                assert all(v <= piece for v in links.values())
                if phase == 0:
                    assert all(s == 0 for s, _ in links) and len(links) == len(cur)  # the root's links, each once
                for _, src, dst, off, ln in cur:
                    have[dst].append((off, off + ln))
                for r in range(n):  # coalesce
                    iv = sorted(have[r])
                    out = []
                    for a, b in iv:
                        if out and a <= out[-1][1]:
                            out[-1] = (out[-1][0], max(out[-1][1], b))
                        else:
                            out.append((a, b))
                    have[r] = out
            assert all(h == [(0, size)] for h in have), (n, size)
            # volume: every byte reaches every non-root rank exactly once
            assert sum(x[4] for x in xs) == size * (n - 1)
