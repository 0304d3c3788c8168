"""Build the HIP shared library in-tree: mindaudio_amd/lib/libmindaudio_amd.so (gfx950 only).

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so is
git-ignored but travels to the GPU box with the repo snapshot.
"""
import glob
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB_DIR = os.path.join(PKG, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libmindaudio_amd.so")
ARCH = "gfx950"


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: cannot build the gfx950 kernels")
    return exe


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def is_stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.inc")) + \
        glob.glob(os.path.join(PKG, "..", "include", "*.h"))
    return any(os.path.getmtime(d) > t for d in deps)


FLAGS = ["-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize"]


def _stamp():
    """What the objects depend on besides their sources: target, flags, compiler (path + version)."""
    exe = _hipcc()
    try:
        ver = subprocess.run([exe, "--version"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT).stdout.decode(errors="replace")
    except OSError:
        ver = "?"
    return "arch=%s\nflags=%s\nhipcc=%s\n%s" % (ARCH, " ".join(FLAGS), exe, ver.strip())


def build(force=False, verbose=False):
    """Compile every .hip under csrc/ into one shared object. Returns its path."""
    obj_dir = os.path.join(LIB_DIR, "obj")
    stamp_path = os.path.join(obj_dir, "build.stamp")
    stamp = _stamp() if shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc") else None
    old = open(stamp_path).read() if os.path.exists(stamp_path) else None
    if stamp is not None and old is not None and old != stamp:
        force = True  # another target / flag set / compiler: every object is stale, whatever its mtime says
    if not force and not is_stale():
        if stamp is not None and old is None and os.path.isdir(obj_dir):
            with open(stamp_path, "w") as fh:  # (objects of a tree built before the stamp existed: same recipe)
                fh.write(stamp)
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    objs = []
    os.makedirs(obj_dir, exist_ok=True)
    procs = []
    headers = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.inc")) + \
        glob.glob(os.path.join(PKG, "..", "include", "*.h"))
    t_hdr = max(os.path.getmtime(h) for h in headers)
    for src in sources():
        obj = os.path.join(obj_dir, os.path.basename(src) + ".o")
        objs.append(obj)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), t_hdr):
            continue  # this object is current: only edited sources (or everything, after a header edit) are recompiled
        # compile to a temporary name and rename on success: an interrupted compile must not leave a truncated object that is newer
        # than its source
        cmd = [_hipcc(), "--offload-arch=" + ARCH] + FLAGS + ["-c", src, "-o", obj + ".tmp"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((src, obj, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, obj, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (src, out.decode(errors="replace")))
        os.replace(obj + ".tmp", obj)
    if stamp is not None:
        with open(stamp_path, "w") as fh:
            fh.write(stamp)
    cmd = [_hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB_PATH + ".tmp"] + objs
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if res.returncode != 0:
        raise RuntimeError("link failed:\n" + res.stdout.decode(errors="replace"))
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
