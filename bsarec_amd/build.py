"""Build the gfx950 shared library in-tree with hipcc (cross-compiles without a GPU).

    python -m bsarec_amd.build          # -> bsarec_amd/libbsarec_hip.so
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "bsarec_hip.hip")
DEPS = [os.path.join(HERE, "csrc", f) for f in ("bsarec_hip.hip", "common.h", "gemm.h", "epilogues.h", "kernels.h", "fused_layer.h", "dw_direct.h", "fused_top.h", "fused_chain.h", "comm.h", "catalogue_shard.h")]
DEPS.append(os.path.join(os.path.dirname(HERE), "include", "bsarec_hip.h"))
DEPS.append(os.path.join(os.path.dirname(HERE), "include", "bsarec_comm.h"))
DEPS.append(os.path.join(os.path.dirname(HERE), "include", "bsarec_shard.h"))
OUT = os.path.join(HERE, "libbsarec_hip.so")


def hipcc_path():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


def is_stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=True):
    if not force and not is_stale():
        return OUT
    # -fno-slp-vectorize: packed f32 VALU (v_pk_add_f32 / v_pk_fma_f32, which the SLP vectoriser forms from adjacent scalar
    # adds / multiplies) issues worse beside MFMAs (MI355X guide); measured A/B on one box, three runs each: 0.1688 -> 0.1680 ms/step
    # -target-feature -packed-fp32-ops (device side; the host pass prints "not a recognized feature" and ignores it): no
    # v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 at all -- the back end forms them from float4 arithmetic even without the SLP
    # vectoriser.  Correctness, not speed: with them, the FrequencyLayer of the <head size 32, full block, x3_products> forward
    # kernel intermittently returned wrong spectra for whole sequences (packed FMAs of waves 0..3 issuing back to back while
    # their SIMD partners run bf16 MFMAs; 2 ... 9 of 10 runs wrong, none of 45 without packed ops: tools/dbg/x3_case.py)
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-munsafe-fp-atomics", "-fno-slp-vectorize",
           "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops", SRC, "-o", OUT]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
    err = "\n".join(l for l in r.stderr.splitlines() if "'-packed-fp32-ops' is not a recognized feature" not in l)
    if err.strip():
        print(err, file=sys.stderr)
    if r.returncode != 0:
        raise subprocess.CalledProcessError(r.returncode, cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
