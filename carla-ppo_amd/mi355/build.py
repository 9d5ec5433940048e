"""Build libmi355_carla.so for gfx950 with hipcc (cross-compiles without a GPU). In-tree output so the
artefact travels with the repo snapshot to the GPU box."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "csrc")
ROOT = os.path.dirname(os.path.dirname(HERE))
OBJ = os.path.join(ROOT, "build", "obj")
LIB = os.path.join(HERE, "libmi355_carla.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-I", os.path.join(ROOT, "include")]
FLAGS += os.environ.get("HIPCC_EXTRA", "").split()        # e.g. -DMI355_PPO_PAD_PROBE (the boundary-cost probe of profiles/r05_ppo.md); never set by the product build


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _newer(a, b):
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def build_variant(tag, extra_flags, verbose=True):
    """A second build of the library with extra compiler flags into build/lib_<tag>.so (objects in build/obj_<tag>): compiler-flag A/B runs load it with MI355_LIB=build/lib_<tag>.so
    next to the product build (tools/ab_env.sh).  Never loaded by the product."""
    global OBJ, LIB, FLAGS
    saved = (OBJ, LIB, FLAGS)
    OBJ = os.path.join(ROOT, "build", "obj_" + tag)
    LIB = os.path.join(ROOT, "build", "lib_%s.so" % tag)
    FLAGS = FLAGS + list(extra_flags)
    try:
        return _build(False, verbose, [])
    finally:
        OBJ, LIB, FLAGS = saved


def build(force=False, verbose=True, asan=False):
    """asan=True: the HOST side of the library with AddressSanitizer (-fsanitize=address is ignored for the gfx950 code objects) into build/asan/ -- the
    checker build of tools/asan_host_check.sh (sanitizer row); never loaded by the product."""
    global OBJ, LIB, FLAGS
    if asan:
        saved = (OBJ, LIB, FLAGS)
        OBJ = os.path.join(ROOT, "build", "obj_asan")
        os.makedirs(os.path.join(ROOT, "build", "asan"), exist_ok=True)
        LIB = os.path.join(ROOT, "build", "asan", "libmi355_carla.so")
        FLAGS = [f if f != "-O3" else "-O1" for f in FLAGS] + ["-g", "-fsanitize=address", "-shared-libsan", "-fno-omit-frame-pointer", "-Wno-option-ignored"]
        try:
            return _build(force, verbose, ["-fsanitize=address", "-shared-libsan"])
        finally:
            OBJ, LIB, FLAGS = saved
    return _build(force, verbose, [])


def _build(force, verbose, link_extra):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    hdr_m = max(os.path.getmtime(os.path.join(CSRC, f)) for f in os.listdir(CSRC) if f.endswith(".hpp"))
    jobs = []
    for s in sources():
        src, obj = os.path.join(CSRC, s), os.path.join(OBJ, s[:-4] + ".o")
        if force or _newer(src, obj) or hdr_m > os.path.getmtime(obj):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        r = subprocess.run([hipcc] + FLAGS + ["-c", src, "-o", obj], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr[-4000:]))
        return src

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        for done in ex.map(cc, jobs):
            if verbose:
                print("[mi355.build] compiled", os.path.basename(done), flush=True)
    objs = [os.path.join(OBJ, s[:-4] + ".o") for s in sources()]
    if force or jobs or not os.path.exists(LIB):
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + link_extra + objs + ["-ldl"], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s" % r.stderr[-4000:])
        if verbose:
            print("[mi355.build] linked", LIB, flush=True)
    return LIB


if __name__ == "__main__":
    if "--variant" in sys.argv:                           # python -m mi355.build --variant <tag> <flag> <flag> ...
        i = sys.argv.index("--variant")
        build_variant(sys.argv[i + 1], sys.argv[i + 2:])
    else:
        build(force="--force" in sys.argv, asan="--asan" in sys.argv)
