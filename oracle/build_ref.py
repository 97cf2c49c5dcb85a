"""Recipe for oracle/_ref/: the reference's OWN hot-path Python, compiled to bytecode, so that it can run where
/root/reference does not exist (the GPU box) — as the checker and as bench.py's `cpu_baseline` (kind "reference").

TEST INFRASTRUCTURE ONLY.  The reference is Python, so "building" it means `py_compile`: every module the shimmed
`ZigMa.forward` needs is compiled FROM WHERE IT LIES under /root/reference into a sourceless `.pyc`
under oracle/_ref/ (same package tree).  No reference source is copied; oracle/_ref/ is listed in .gitignore (it stays
out of history) but not in .gpurunignore (it travels with the snapshot like the built .so).  oracle/ref_shim.py imports
from /root/reference when it exists and from oracle/_ref/ otherwise.

    python -m oracle.build_ref          # in the build container; __graft_entry__.build() calls it too
"""
import os
import py_compile
import sys

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")

# module files on the denoiser-forward path (model_zigma.py:911-990 -> mamba_simple.py:274-444 ->
# selective_scan_interface.py:86-152,296-365 -> causal_conv1d_interface.py:49-65) + the scan-order tables
MODULES = [
    "model_zigma.py",
    "utils/utils_zigzag.py",
    "dis_mamba/mamba_ssm/modules/mamba_simple.py",
    "dis_mamba/mamba_ssm/ops/selective_scan_interface.py",
    "dis_causal_conv1d/causal_conv1d/__init__.py",
    "dis_causal_conv1d/causal_conv1d/causal_conv1d_interface.py",
]


def build(verbose=True):
    """Returns OUT, or None when /root/reference is absent (then an already shipped oracle/_ref/ is used as is)."""
    if not os.path.isdir(REF):
        return OUT if os.path.isdir(OUT) else None
    for rel in MODULES:
        src = os.path.join(REF, rel)
        dst = os.path.join(OUT, rel[:-3] + ".pyc")
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        py_compile.compile(src, cfile=dst, dfile=rel, doraise=True,
                           invalidation_mode=py_compile.PycInvalidationMode.UNCHECKED_HASH)
    with open(os.path.join(OUT, "PYTHON_TAG"), "w") as fh:      # bytecode is tied to the interpreter version
        fh.write(sys.implementation.cache_tag + "\n")
    if verbose:
        print(f"oracle/_ref: {len(MODULES)} reference modules compiled to bytecode ({sys.implementation.cache_tag})")
    return OUT


def available():
    """True when the reference can be imported here (from its checkout or from the compiled copy)."""
    if os.path.isdir(REF):
        return True
    tag = os.path.join(OUT, "PYTHON_TAG")
    return os.path.exists(tag) and open(tag).read().strip() == sys.implementation.cache_tag


if __name__ == "__main__":
    print(build())
