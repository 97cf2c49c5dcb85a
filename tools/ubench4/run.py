"""Build + run tools/ubench4/tr_probe.hip: the lane/element mapping of ds_read_b64_tr_b16 for three address patterns."""
import ctypes, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
src, lib = os.path.join(HERE, "tr_probe.hip"), os.path.join(HERE, "libtrprobe.so")
if "--no-build" not in sys.argv:
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", src, "-o", lib], check=True)
import torch
if not torch.cuda.is_available():
    print("built", lib); sys.exit(0)
L = ctypes.CDLL(lib)
L.tr_probe.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
for pat in (0, 1, 2):
    out = torch.zeros(256, dtype=torch.int16, device="cuda")
    assert L.tr_probe(out.data_ptr(), pat, torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.synchronize()
    o = out.cpu().view(64, 4).tolist()
    print("pattern", pat)
    for l in (0, 1, 2, 3, 4, 15, 16, 17, 31, 32, 48, 63):
        print("  lane", l, o[l])
