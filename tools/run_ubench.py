"""Build + run tools/ubench.hip on the GPU; prints cycles per wave-instruction per SIMD for each mode."""
import ctypes, json, os, subprocess, sys
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libubench.so")


def build():
    src = os.path.join(HERE, "ubench.hip")
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                        "-ffp-contract=fast", src, "-o", LIB], check=True)
    return LIB


def main():
    build()
    if not torch.cuda.is_available():
        print("built; no GPU")
        return
    L = ctypes.CDLL(LIB)
    L.ubench_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    out = torch.empty(256 * 8 * 256 * 2, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    names = {0: "exp2 x8", 1: "pk_fma x8", 2: "fma x8", 3: "mix 8 exp + 16 pk_fma (one wave)",
             4: "split waves: even exp x8 / odd pk_fma x16", 5: "soft exp2 (pk) x16 values", 6: "log2 x8", 7: "rcp x8",
             8: "mix 8 exp + 16 scalar fma", 9: "mul_dpp row_newbcast x8", 10: "fmac_dpp row_newbcast x8",
             11: "scan core replica: 2 steps x 4 states (40 VALU incl 8 exp)", 12: "exp_f16 x8", 13: "rcp_f16 x8",
             14: "pk_mul_f16 x8"}
    clk = 2.4e9
    res = {}
    iters = 4000
    for wpc in (1, 2, 4, 5):           # 256-thread blocks per CU -> waves per SIMD
        blocks = 256 * wpc
        for mode in names:
            for _ in range(2):
                L.ubench_launch(mode, blocks, iters, out.data_ptr(), st)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                L.ubench_launch(mode, blocks, iters, out.data_ptr(), st)
            e1.record()
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1) * 1e-3 / 5
            # cycles per loop iteration per SIMD (wpc waves share a SIMD)
            cyc_iter = t * clk / iters / wpc
            res[f"w{wpc}_m{mode}"] = dict(name=names[mode], waves_per_simd=wpc, time_ms=t * 1e3, cycles_per_iter_per_wave=cyc_iter)
            print(f"waves/SIMD={wpc}  {names[mode]:45s} {t*1e3:8.3f} ms  {cyc_iter:8.1f} cyc/iter/wave (@2.4GHz)")
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(res, open("gpurun_out/ubench.json", "w"), indent=1)


if __name__ == "__main__":
    main()
