"""Build libduodiff.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m duodiff_amd.build [--force]

One object per HIP translation unit (compiled in parallel), linked into
``duodiff_amd/libduodiff.so``.  The built library travels with the tree; nothing is
JIT-compiled at run time.
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
REPO = PKG.parent
CSRC = PKG / "csrc"
OBJ = REPO / "build" / "obj"
LIB = PKG / "libduodiff.so"
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++20", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
         "-fno-gpu-rdc", f"-I{REPO / 'include'}"]


def hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found: the MI355X engine cannot be built (no CPU fallback exists)")


def sources():
    return sorted(CSRC.glob("*.hip"))


def source_id(extra_flags=()) -> str:
    """Build id of the library: hash of every source and header it is compiled from (+ non-default flags).  Exported as
    dd_build_id(); profiles/rNN/pmc_*.json record it, and bench.py quotes a profile's counters only for the build they
    were collected on."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted([*CSRC.glob("*.hip"), *CSRC.glob("*.h"), *(REPO / "include").glob("*.h")]):
        h.update(f.name.encode())
        h.update(f.read_bytes())
    for fl in extra_flags:
        h.update(fl.encode())
    return h.hexdigest()[:16]


def _stale(out: Path, deps) -> bool:
    if not out.exists():
        return True
    t = out.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps)


def build_library(force: bool = False, verbose: bool = False) -> Path:
    cc = hipcc()
    OBJ.mkdir(parents=True, exist_ok=True)
    headers = list(CSRC.glob("*.h")) + list((REPO / "include").glob("*.h"))
    extra = os.environ.get("DD_EXTRA_HIPCC_FLAGS", "").split()
    bid = source_id(extra)
    id_file = OBJ / "build_id.txt"
    id_stale = not id_file.exists() or id_file.read_text().strip() != bid
    jobs = []
    for src in sources():
        obj = OBJ / (src.stem + ".o")
        if force or _stale(obj, [src, *headers]) or (id_stale and src.stem == "capi"):   # capi.hip carries the id
            jobs.append((src, obj))

    def compile_one(job):
        src, obj = job
        cmd = [cc, *FLAGS, *extra, f'-DDD_BUILD_ID="{bid}"', "-c", str(src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src.name}:\n{r.stdout}\n{r.stderr}")
        return r.stderr

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            for warn in ex.map(compile_one, jobs):
                if verbose and warn.strip():
                    print(warn)
    objs = [OBJ / (s.stem + ".o") for s in sources()]
    if force or jobs or _stale(LIB, objs):
        cmd = [cc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc", *map(str, objs), "-o", str(LIB)]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    id_file.write_text(bid + "\n")
    return LIB


if __name__ == "__main__":
    path = build_library(force="--force" in sys.argv, verbose=True)
    print(f"built {path} ({path.stat().st_size / 1e6:.1f} MB)")
