"""Probe (GPU box): how fast does a tmpfs file reach the device when its mapping is registered with HIP instead of being copied
through pinned staging buffers?  python tools/hostreg_probe.py [GB=8]"""
import ctypes as C
import mmap
import os
import sys
import time

import torch

gb = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
n = int(gb * (1 << 30))
path = "/dev/shm/k4_hostreg_probe.bin"
t0 = time.time()
with open(path, "wb") as f:
    chunk = os.urandom(1 << 20) * 64
    w = 0
    while w < n:
        f.write(chunk[: min(len(chunk), n - w)])
        w += min(len(chunk), n - w)
print("file written in %.1fs" % (time.time() - t0), flush=True)
hip = C.CDLL("libamdhip64.so")
torch.cuda.init()
d = torch.empty(n, dtype=torch.uint8, device="cuda")
for flags, name in ((os.O_RDWR, "MAP_SHARED rw"), (os.O_RDONLY, "MAP_PRIVATE ro")):
    fd = os.open(path, flags)
    if flags == os.O_RDWR:
        m = mmap.mmap(fd, n, mmap.MAP_SHARED, mmap.PROT_READ | mmap.PROT_WRITE)
    else:
        m = mmap.mmap(fd, n, mmap.MAP_PRIVATE, mmap.PROT_READ)
    addr = C.addressof(C.c_char.from_buffer(m)) if flags == os.O_RDWR else None
    if addr is None:
        buf = (C.c_char * n).from_buffer_copy(b"") if False else None
        # read-only mappings cannot be exported through from_buffer: take the address via ctypes' mmap of libc
        libc = C.CDLL("libc.so.6")
        libc.mmap.restype = C.c_void_p
        libc.mmap.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_long]
        addr = libc.mmap(None, n, mmap.PROT_READ, mmap.MAP_PRIVATE, fd, 0)
    for hflag, hname in ((0, "default"), (0x100, "readonly-flag")):
        t0 = time.time()
        rc = hip.hipHostRegister(C.c_void_p(addr), C.c_size_t(n), C.c_uint(hflag))
        t_reg = time.time() - t0
        if rc != 0:
            print(name, hname, "hipHostRegister rc", rc, "in %.2fs" % t_reg, flush=True)
            continue
        torch.cuda.synchronize()
        t0 = time.time()
        rc2 = hip.hipMemcpy(C.c_void_p(d.data_ptr()), C.c_void_p(addr), C.c_size_t(n), C.c_int(1))
        torch.cuda.synchronize()
        t_cp = time.time() - t0
        t0 = time.time()
        hip.hipHostUnregister(C.c_void_p(addr))
        t_un = time.time() - t0
        print("%s %s: register %.2fs, copy %.2fs (%.1f GB/s, rc %d), unregister %.2fs -> %.1f GB/s overall" %
              (name, hname, t_reg, t_cp, n / t_cp / 1e9, rc2, t_un, n / (t_reg + t_cp) / 1e9), flush=True)
        break
    os.close(fd)
os.remove(path)
