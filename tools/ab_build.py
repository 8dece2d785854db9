"""A/B builds of ONE kernel file: tools/ab_build.py k_implicit_mfma NAME -DMACRO [-DMACRO2 ...] compiles csrc/<file>.hip with
the extra flags and links it with the product's other objects into mici_amd/lib/ab_<NAME>.so (gitignored; travels to the GPU
box).  Run with MICI_AMD_LIB=mici_amd/lib/ab_<NAME>.so python bench.py --config c3 ..."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mici_amd import build as mb  # noqa: E402


def main():
    stem, name, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
    src = os.path.join(mb.CSRC, stem + ".hip")
    obj = os.path.join(mb.OBJ, f"ab_{name}_{stem}.o")
    cmd = [mb.hipcc(), *mb.FLAGS, *mb.EXTRA_FLAGS.get(stem + ".hip", []), *extra, "-c", src, "-o", obj]
    subprocess.run(cmd, check=True, capture_output=True)
    others = [os.path.join(mb.OBJ, f) for f in sorted(os.listdir(mb.OBJ))
              if f.endswith(".o") and not f.startswith("ab_") and f != stem + ".o"]
    lib = os.path.join(mb.HERE, "lib", f"ab_{name}.so")
    subprocess.run([mb.hipcc(), f"--offload-arch={mb.ARCH}", "-shared", "-fPIC", obj, *others, "-o", lib, "-ldl"], check=True)
    os.remove(obj)
    print(lib)


if __name__ == "__main__":
    main()
