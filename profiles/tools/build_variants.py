"""Alternative builds of libpram_hip.so for A/B timing on one box (PRAM_HIP_LIB selects one at import time; never used in production).
    python profiles/tools/build_variants.py TAG:file.hip[+file2.hip]:FLAG,FLAG ...      ->  pram_amd/csrc/variants/libpram_hip_TAG.so
Every other object is the in-tree one (python -m pram_amd.build first).  Example:
    python profiles/tools/build_variants.py scalar:attention_x3.hip:-DAX_SCALAR=1,-fno-slp-vectorize
"""
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from pram_amd import build as B  # noqa: E402

OUT = B.CSRC / "variants"


def one(spec):
    tag, files, flags = (spec.split(":") + ["", ""])[:3]
    flags = [f for f in flags.split(",") if f]
    OUT.mkdir(exist_ok=True)
    repl = {}
    for f in files.split("+"):
        src = B.CSRC / f
        obj = OUT / f"{src.stem}_{tag}.o"
        cmd = [B.HIPCC, *B.flags_for(src), *flags, "-c", str(src), "-o", str(obj)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError(f"{tag}: {r.stderr}")
        repl[src.with_suffix(".o")] = obj
    objs = [repl.get(s.with_suffix(".o"), s.with_suffix(".o")) for s in B.sources()]
    lib = OUT / f"libpram_hip_{tag}.so"
    r = subprocess.run([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *map(str, objs), "-o", str(lib)], capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError(f"{tag}: link: {r.stderr}")
    return lib


if __name__ == "__main__":
    B.build(verbose=False)
    with ThreadPoolExecutor(max_workers=8) as ex:
        for lib in ex.map(one, sys.argv[1:]):
            print(lib)
