"""Builds libevo_mi355x.so (the C-ABI of include/evo_mi355x.h) from evo_amd/csrc/*.hip with hipcc for gfx950.

In-tree build: the .so lands in evo_amd/_lib/ so it travels with the repo snapshot to the GPU box
(it is git-ignored).  hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

import os
import shutil
import subprocess
from pathlib import Path

ROOT = Path(__file__).resolve().parent
CSRC = ROOT / "csrc"
LIBDIR = ROOT / "_lib"
# A/B builds: EVO_AMD_LIBNAME picks another file name in _lib/, EVO_AMD_HIPCC_FLAGS appends compiler flags
LIBNAME = os.environ.get("EVO_AMD_LIBNAME", "libevo_mi355x.so")
ARCH = "gfx950"

# every symbol include/evo_mi355x.h declares
EXPORTS = [
    "evo_abi_version", "evo_embed_bf16", "evo_rmsnorm_bf16", "evo_hyena_seg_state", "evo_hyena_carry_scan", "evo_hyena_carry_add",
    "evo_hyena_apply", "evo_hyena_step", "evo_rope_qk_bf16", "evo_attn_fwd_causal_bf16", "evo_attn_decode_bf16",
    "evo_linear_small_m_bf16", "evo_mlp_gate_small_m_bf16", "evo_norm_mlp_gate_small_m_bf16", "evo_norm_linear_small_m_bf16", "evo_hyena_decode_fused_small_m", "evo_linear_mfma_bf16", "evo_mlp_gate_mfma_bf16", "evo_linear_xblk_mfma_bf16", "evo_hyena_ct", "evo_linear_t_mfma_bf16", "evo_rmsnorm_rows_bf16", "evo_gelu_gate_bf16",
    "evo_logprob_entropy", "evo_unembed_logprob_bf16", "evo_rope_append_decode_bf16",
    "evo_linear_mfma_nf_bf16", "evo_linear_xblk_mfma_nf_bf16", "evo_mlp_gate_mfma_nf_bf16", "evo_linear_t_mfma_nf_bf16", "evo_rms_finalize_f32",
    "evo_probe_copy_f4", "evo_probe_mfma_bf16",
]


def lib_path() -> Path:
    return LIBDIR / LIBNAME


def sources():
    return sorted(CSRC.glob("*.hip"))


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build libevo_mi355x.so")


def is_stale() -> bool:
    out = lib_path()
    if not out.exists():
        return True
    deps = list(sources()) + list(CSRC.glob("*.h")) + [ROOT.parent / "include" / "evo_mi355x.h"]
    newest = max(p.stat().st_mtime for p in deps if p.exists())
    return newest > out.stat().st_mtime


# per-file compiler flags.  attn_w64.hip: its softmax streams are scalar fp32 on purpose (beside MFMAs a packed v_pk_* instruction costs
# more than the two it replaces), so the SLP vectoriser stays off for that file
FILE_FLAGS = {"attn_w64.hip": ["-fno-slp-vectorize", "-Wno-unused-value"]}


def _compile_one(args):
    src, obj, verbose = args
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wno-inline-asm"] \
        + os.environ.get("EVO_AMD_HIPCC_FLAGS", "").split() + FILE_FLAGS.get(src.name, []) + ["-c", str(src), "-o", str(obj)]
    if verbose:
        print(" ".join(cmd))
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src.name}:\n" + proc.stdout + proc.stderr)
    return obj


def build(force: bool = False, verbose: bool = False) -> Path:
    """Compile every HIP source (one object per file, in parallel) and link them into one shared library.  Returns the library path."""
    out = lib_path()
    if not force and not is_stale():
        return out
    LIBDIR.mkdir(exist_ok=True)
    objdir = LIBDIR / ("obj%d" % os.getpid())
    objdir.mkdir(exist_ok=True)
    try:
        from concurrent.futures import ThreadPoolExecutor
        jobs = [(s, objdir / (s.stem + ".o"), verbose) for s in sources()]
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            objs = list(ex.map(_compile_one, jobs))
        tmp = out.with_suffix(".so.tmp%d" % os.getpid())
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-fno-gpu-rdc", "-o", str(tmp)] + [str(o) for o in objs]
        if verbose:
            print(" ".join(cmd))
        proc = subprocess.run(cmd, capture_output=True, text=True)
        if proc.returncode != 0:
            raise RuntimeError("hipcc (link) failed:\n" + proc.stdout + proc.stderr)
        os.replace(tmp, out)
    finally:
        shutil.rmtree(objdir, ignore_errors=True)
    return out


if __name__ == "__main__":
    print(build(force=True, verbose=True))
