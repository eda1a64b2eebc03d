#!/usr/bin/env python3
"""GPU box diagnostic: builds the library with -DSP_STAMPS and prints where a tile's cycles go in k_cc_sparse."""
import ctypes, os, subprocess, sys, glob
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib = "/tmp/libstamps.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-mllvm", "-amdgpu-atomic-optimizer-strategy=DPP", "-fPIC", "-shared", "-DSP_STAMPS",
                       "-o", lib] + sorted(glob.glob(os.path.join(ROOT, "pymasc_amd/csrc/*.hip"))))
os.environ["PYMASC_AMD_LIB"] = lib
os.environ["PMX_CC_EVENTS"] = "0"       # the window kernel on every tile (otherwise it only sees what the event kernel flags)
os.environ["PMX_AUTOCORR_FORK"] = "0"
import torch
from pymasc_amd import ffi, synth
mode = sys.argv[1] if len(sys.argv) > 1 else "both"
with_m = mode == "both"
ctx = ffi.Context(0)
dev = torch.device("cuda", 0)
S, L = 1000, 36
vecs = [synth.make_chromosome(ctx, dev, n, ln, S, L, 0xC0FFEE + i, with_m=with_m) for i, (n, ln) in enumerate(synth.HG38)]
out = torch.zeros((len(vecs), ffi.PMX_NROWS, S + 1), dtype=torch.int64, device=dev)
args = ([v.F.data_ptr() for v in vecs], [v.R.data_ptr() for v in vecs], [v.M.data_ptr() for v in vecs] if with_m else None,
        [v.nbits for v in vecs], S, L, 0, [out[i].data_ptr() for i in range(len(vecs))])
for _ in range(3):
    ctx.cc_batch_dev(*args)
ctx.sync()
ntiles = sum((v.nbits + 32767) // 32768 for v in vecs)
per_cu = 3 if with_m else 4
nwg = min(256 * per_cu, ntiles)
tpw = -(-ntiles // nwg); nwg = -(-ntiles // tpw)
# the autocorr launch reuses the slab afterwards, so re-run the cc kernel alone to read its stamps
if with_m:
    L_ = ffi.load_library()
ctx.cc_batch_dev(args[0], args[1], None if not with_m else args[2], args[3], S, L, 0, args[7])
ctx.sync()
off = (nwg + len(vecs)) * 5 * 1024 * 4
NS = 12
buf = np.zeros(nwg * 4 * NS, dtype=np.uint64)
Lb = ffi.load_library()
Lb.pmx_debug_read_slab.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64]
rc = Lb.pmx_debug_read_slab(ctx._h, off, buf.ctypes.data, buf.nbytes)
assert rc == 0
a = buf.reshape(nwg, 4, NS).astype(np.float64)
tot = a.sum(axis=2).mean()
print(f"mode={mode} nwg={nwg} tiles/wg={tpw} cycles per wave lifetime={tot:.0f}  per tile={tot / tpw:.0f}")
print("  stamp -> phase share")
lab = {0: "B0 barrier wait", 1: "fold/convert check", 2: "tile_store + decimate", 3: "emit (reserve + records)", 4: "prefetch issue",
       5: "B1 barrier wait", 6: "process (pads + F + R loops)", 7: "loop tail / job change",
       8: "  emit: reserve F (atomic)", 9: "  emit: reserve R (atomic)", 10: "  emit: positions F", 11: "-"}
lab[3] = "emit: positions R + cursor reset"
for i in range(NS):
    print(f"  {lab[i]:32s} {a[:, :, i].mean() / tpw:9.0f} cyc/tile  {100 * a[:, :, i].sum() / a.sum():5.1f} %")
