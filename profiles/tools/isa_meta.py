"""Per-kernel resource usage (VGPRs, AGPRs, SGPRs, scratch bytes, LDS bytes, spills) of the gfx950 code objects inside the
in-tree object files, read from the code-object metadata — no GPU needed.
    python profiles/tools/isa_meta.py [pram_amd/csrc/linear.o ...]      (default: every .o under pram_amd/csrc)
"""
import glob, os, re, subprocess, sys, tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
files = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "pram_amd", "csrc", "*.o")))
run = lambda *a: subprocess.run(a, capture_output=True, text=True)
with tempfile.TemporaryDirectory() as tmp:
    for f in files:
        b = os.path.basename(f)[:-2]
        fat, co = os.path.join(tmp, b + ".fatbin"), os.path.join(tmp, b + ".co")
        if run(f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", f).returncode:
            continue
        if run(f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
               f"--output={co}").returncode:
            continue
        notes = run(f"{LLVM}/llvm-readelf", "--notes", co).stdout
        dis = run(f"{LLVM}/llvm-objdump", "-d", co).stdout
        print(f"== {b}: static v_mfma {dis.count('v_mfma')}, scratch_ instructions {dis.count('scratch_')}")
        for blk in notes.split("- .agpr_count:")[1:]:
            g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
            name = run("c++filt", g("name")).stdout.strip()
            name = re.sub(r"\(anonymous namespace\)::", "", name)
            name = re.sub(r"\(.*", "", name)
            if name.startswith("_ZN"):
                mm = re.match(r"_ZN12_GLOBAL__N_1\d+([a-z0-9_]+?_kernel)I(.*?)EEv", name)
                if mm:
                    name = mm.group(1) + "<" + ",".join(re.findall(r"L[ib](\d+)E", mm.group(2))) + ">"
            print(f"  {name[:70]:70s} vgpr {g('vgpr_count'):>3s} agpr {blk.split()[0]:>3s} sgpr {g('sgpr_count'):>3s} "
                  f"scratch {g('private_segment_fixed_size'):>4s} lds {g('group_segment_fixed_size'):>6s} vspill {g('vgpr_spill_count'):>3s}")
