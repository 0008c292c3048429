"""Build libpram_hip.so (gfx950) in-tree with hipcc.  ``python -m pram_amd.build [--force]``."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

CSRC = Path(__file__).resolve().parent / "csrc"
LIB = CSRC / "libpram_hip.so"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -ffp-contract=off: the boundary promises the reference's fp32 op order where it is cheap to keep
# (coordinate maps, rotary, Sinkhorn scaling); hot VALU loops call fmaf explicitly.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wall", "-Wno-unused-function"]
# per-file additions to FLAGS (measured A/Bs: profiles/r05_*; MI355X_MICROARCH.md: packed fp32 VALU beside MFMAs is an anti-lever)
FILE_FLAGS: dict[str, list[str]] = {
    # the soft-max of the attention kernel is scalar fp32 on purpose: SLP would re-pack it into v_pk_* (+18 clocks beside an MFMA)
    "attention_x3.hip": ["-fno-slp-vectorize"],
}


def flags_for(src: Path) -> list[str]:
    return FLAGS + FILE_FLAGS.get(src.name, [])


def sources():
    return sorted(CSRC.glob("*.hip"))


def _stale(obj: Path, src: Path) -> bool:
    if not obj.exists():
        return True
    t = obj.stat().st_mtime
    deps = [src] + list(CSRC.glob("*.h")) + [CSRC.parent.parent / "include" / "pram_hip.h"]
    return any(d.stat().st_mtime > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> Path:
    objs = []
    jobs = []
    for src in sources():
        obj = src.with_suffix(".o")
        objs.append(obj)
        if force or _stale(obj, src):
            jobs.append((src, obj))

    def compile_one(job):
        src, obj = job
        cmd = [HIPCC, *flags_for(src), "-c", str(src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src.name}:\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(compile_one, jobs))
    if jobs or not LIB.exists():
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *map(str, objs), "-o", str(LIB)]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
