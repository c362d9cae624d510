#!/usr/bin/env python3
"""Build a development variant of the library next to the product one:

    python tools/build_variant.py NAME [--csrc DIR] [more hipcc flags]

compiles every *.hip of csrc (or of DIR: e.g. the csrc directory of a `git worktree` of an earlier commit, for a same-box
A/B of two kernel versions) with the extra flags into duodiff_amd/libduodiff_NAME.so (objects under build/obj_NAME).
Select it at run time with DUODIFF_LIB=duodiff_amd/libduodiff_NAME.so.  The product build (python -m duodiff_amd.build)
holds no ablation or instrumentation code paths.
"""
import subprocess
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
from duodiff_amd.build import FLAGS, ARCH, hipcc, sources  # noqa: E402


def main():
    name, extra = sys.argv[1], sys.argv[2:]
    srcs = sources()
    if "--csrc" in extra:
        i = extra.index("--csrc")
        srcs = sorted(Path(extra[i + 1]).resolve().glob("*.hip"))
        del extra[i:i + 2]
    obj = REPO / "build" / f"obj_{name}"
    obj.mkdir(parents=True, exist_ok=True)
    cc = hipcc()
    objs = []
    for src in srcs:
        o = obj / (src.stem + ".o")
        subprocess.run([cc, *FLAGS, *extra, "-c", str(src), "-o", str(o)], check=True)
        objs.append(str(o))
    lib = REPO / "duodiff_amd" / f"libduodiff_{name}.so"
    subprocess.run([cc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc", *objs, "-o", str(lib)], check=True)
    print("built", lib)


if __name__ == "__main__":
    main()
