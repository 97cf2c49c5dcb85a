"""Build libzigma_hip.so (gfx950) in-tree with hipcc.  `python -m zigma_amd.build [--force]`."""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "lib", "libzigma_hip.so")
SOURCES = ["api.hip", "selective_scan.hip", "scan_tok_bf16.hip", "scan_tok_f16.hip", "scan_tok_f32.hip",
           "causal_conv1d.hip", "add_norm.hip", "dt_proj.hip", "scan_bwd.hip", "conv_bwd.hip", "norm_bwd.hip", "cross_attn.hip", "cross_attn_bwd.hip", "x_proj.hip", "linear.hip", "linear4w.hip", "linear_ws.hip", "linear_sm.hip", "conv_x_proj.hip", "glue_bwd.hip", "embed.hip", "skinny_linear.hip", "calib.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-fno-slp-vectorize",
         "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]


# per-source flags.  cross_attn.hip: MFMA results in VGPRs — its accumulators are consumed by VALU code right away (softmax, normalisation);
# left to the default (AGPR destinations in a kernel with registers to spare) every result costs a v_accvgpr_read_b32
# (the same switch on conv_x_proj.hip and cross_attn_bwd.hip removes their AGPR moves too and changes nothing measurable: 17.21-17.28 vs
# 17.23-17.26 ms per forward, 97.5-98.2 vs 96.8-98.0 ms per training step, tools/fwd_vgpr_mfma_ab.sh)
_VGPR_MFMA = ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]
# scan_tok2_kernel loads its bf16 rows with inline-asm `buffer_load_short_d16_hi` (csrc/scan_tok2.inc): the compiler does not see those writes, so a spill or
# a live-range split of the destination registers between the load and the tile's fence would silently lose a row.  The two TUs that instantiate
# the kernel are compiled with the resource-usage remarks on, and the build FAILS if any instantiation uses scratch memory or spills a register.
_RES = ["-Rpass-analysis=kernel-resource-usage"]
SOURCE_FLAGS = {"cross_attn.hip": _VGPR_MFMA, "scan_tok_bf16.hip": _RES, "scan_tok_f16.hip": _RES}
NO_SCRATCH_KERNELS = ("scan_tok2_kernel",)


def check_no_scratch(remarks, src):
    """parse -Rpass-analysis=kernel-resource-usage output: every NO_SCRATCH_KERNELS instantiation must report ScratchSize 0 and no spills"""
    import re
    bad, seen, name = [], 0, None
    for ln in remarks.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", ln)
        if m:
            name = m.group(1) if any(k in m.group(1) for k in NO_SCRATCH_KERNELS) else None
            seen += name is not None
            continue
        m = re.search(r"remark:\s+(ScratchSize \[bytes/lane\]|SGPRs Spill|VGPRs Spill): (\d+)", ln)
        if m and name and int(m.group(2)) != 0:
            bad.append((name, m.group(1), int(m.group(2))))
    if bad or not seen:
        raise RuntimeError(f"{src}: scan_tok2_kernel must not spill (inline-asm d16_hi row loads are invisible to the register allocator): "
                           f"{bad if bad else 'no instantiation found in the remarks'}")
    return seen


def _digest(paths, extra=()):
    h = hashlib.sha256()
    for pth in sorted(paths):
        h.update(os.path.basename(pth).encode())
        h.update(open(pth, "rb").read())
    for e in extra:
        h.update(str(e).encode())
    return h.hexdigest()


def build(force=False, verbose=True, sources=None, lib=None, extra_flags=()):
    """Compile every .hip of csrc/ and link the shared library.  Staleness is decided by a CONTENT hash of the
    sources + flags stored next to the library (mtimes do not survive the snapshot copy to the GPU box, and a
    library that already matches its sources must not be rebuilt there)."""
    sources = sources or SOURCES
    lib = lib or LIB
    obj_dir = OBJ if lib == LIB else OBJ + "_" + os.path.splitext(os.path.basename(lib))[0]      # (A/B libraries of tools/ keep their own objects)
    os.makedirs(obj_dir, exist_ok=True)
    os.makedirs(os.path.dirname(lib), exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))]
    headers.append(os.path.join(ROOT, "include", "zigma_hip.h"))
    stamp = lib + ".srchash"
    want = _digest([os.path.join(CSRC, s) for s in sources] + headers, extra=list(FLAGS) + list(extra_flags) + [str(sorted(SOURCE_FLAGS.items()))])
    if not force and os.path.exists(lib) and os.path.exists(stamp) and open(stamp).read().strip() == want:
        if verbose:
            print(f"{lib} is up to date (source hash {want[:12]})")
        return lib

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    objs, jobs = [], []
    for src in sources:
        s_path = os.path.join(CSRC, src)
        o_path = os.path.join(obj_dir, src.replace(".hip", ".o"))
        o_stamp = o_path + ".srchash"
        src_flags = SOURCE_FLAGS.get(src, [])
        o_want = _digest([s_path] + headers, extra=list(FLAGS) + list(extra_flags) + src_flags)
        objs.append(o_path)
        if force or not os.path.exists(o_path) or not os.path.exists(o_stamp) or open(o_stamp).read().strip() != o_want:
            jobs.append(([HIPCC, *FLAGS, *extra_flags, *src_flags, "-c", s_path, "-o", o_path], o_stamp, o_want))

    def compile_one(job):
        cmd, o_stamp, o_want = job
        if _RES[0] in cmd:
            if verbose:
                print(" ".join(cmd), flush=True)
            r = subprocess.run(cmd, check=True, capture_output=True, text=True)
            n = check_no_scratch(r.stderr, cmd[-3])
            if verbose:
                print(f"  {os.path.basename(cmd[-3])}: {n} scan_tok2_kernel instantiations, no scratch, no spills", flush=True)
        else:
            run(cmd)
        open(o_stamp, "w").write(o_want)

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(compile_one, jobs))
    run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", lib])
    open(stamp, "w").write(want)
    return lib


if __name__ == "__main__":
    build(force="--force" in sys.argv)
