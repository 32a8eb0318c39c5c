"""TEST INFRASTRUCTURE: pack the reference's Python tree for the GPU box.

/root/reference does not exist on the MI355X box, but tests/test_gpu_reference_unchanged.py has to run the reference's OWN,
unchanged lib/net + lib/rpn + lib/utils Python on the HIP kernels there.  Like oracle/_ref/libprcnn_ref.so (the reference's
native sources compiled for the host), this is a BUILT ARTEFACT of the checker: one archive under oracle/_ref/ (git-ignored,
never committed, not gpurun-ignored so that it travels with the snapshot).  Nothing in the product reads it; the test unpacks
it into a temporary directory and sets sys.path exactly as tools/_init_path.py:1-4 does.  Staging is an explicit step
(`__graft_entry__.build()` runs it in the build container; `python -m oracle.stage_reference`), not a side effect of oracle.build().

    python -m oracle.stage_reference            # -> oracle/_ref/reference_py.tar.gz
"""
import io
import os
import tarfile

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE = os.environ.get("PRCNN_REFERENCE", "/root/reference")
ARCHIVE = os.path.join(HERE, "_ref", "reference_py.tar.gz")
# what the network-level route and the end-to-end run of tools/eval_rcnn.py need: the library package, the config files, the
# scripts of tools/ with their two helper packages (train_utils: checkpoint save / load; kitti_object_eval_python: the AP evaluator)
WANTED = (("lib", (".py",)), ("tools/cfgs", (".yaml",)), ("tools", (".py",)), ("tools/train_utils", (".py",)),
          ("tools/kitti_object_eval_python", (".py",)))


def have_reference():
    return os.path.isdir(os.path.join(REFERENCE, "lib", "net"))


def members():
    out = []
    for sub, suffixes in WANTED:
        top = os.path.join(REFERENCE, sub)
        for d, dirs, files in os.walk(top):
            dirs[:] = sorted(x for x in dirs if x != "__pycache__" and sub == "lib")      # only lib/ is taken recursively
            for f in sorted(files):
                if f.endswith(suffixes):
                    out.append(os.path.relpath(os.path.join(d, f), REFERENCE))
    return sorted(set(out))


def stage(verbose=False):
    """write the archive (deterministic: sorted members, zeroed times / owners); -> path"""
    if not have_reference():
        raise RuntimeError("reference checkout absent at %s" % REFERENCE)
    os.makedirs(os.path.dirname(ARCHIVE), exist_ok=True)
    buf = io.BytesIO()
    with tarfile.open(fileobj=buf, mode="w") as tar:
        for rel in members():
            info = tarfile.TarInfo(rel)
            with open(os.path.join(REFERENCE, rel), "rb") as f:
                data = f.read()
            info.size, info.mtime, info.mode = len(data), 0, 0o444
            tar.addfile(info, io.BytesIO(data))
    import gzip
    tmp = ARCHIVE + ".tmp.%d" % os.getpid()
    with open(tmp, "wb") as raw, gzip.GzipFile(fileobj=raw, mode="wb", mtime=0) as gz:
        gz.write(buf.getvalue())
    os.replace(tmp, ARCHIVE)
    if verbose:
        print("staged %d files -> %s (%d bytes)" % (len(members()), ARCHIVE, os.path.getsize(ARCHIVE)))
    return ARCHIVE


_UNPACKED = None        # one unpacked copy per process: the reference's modules can only be imported from ONE place


def locate(tmpdir=None):
    """-> a directory holding the reference tree (lib/, tools/cfgs/): $PRCNN_REFERENCE or /root/reference when present, else the
    staged archive unpacked into `tmpdir` (the first caller's; later callers of the same process get that directory again, whatever
    they pass: `lib.*` is importable from one location only); None when neither exists"""
    global _UNPACKED
    if have_reference():
        return REFERENCE
    if _UNPACKED is not None and os.path.isdir(os.path.join(_UNPACKED, "lib", "net")):
        return _UNPACKED
    if os.path.exists(ARCHIVE) and tmpdir is not None:
        with tarfile.open(ARCHIVE, "r:gz") as tar:
            tar.extractall(tmpdir, filter="data")      # plain files under tmpdir only: no links, devices, absolute or .. paths
        _UNPACKED = str(tmpdir)
        return _UNPACKED
    return None


def writable_copy(dest):
    """the same tree as a WRITABLE copy under `dest` (tools/eval_rcnn.py writes next to itself: `cp *.py backup_files/`, `../data`,
    `../output`): unpacked from the archive, or copied file by file from the read-only checkout; -> dest, or None"""
    dest = str(dest)
    if have_reference():
        import shutil
        for rel in members():
            os.makedirs(os.path.dirname(os.path.join(dest, rel)), exist_ok=True)
            shutil.copyfile(os.path.join(REFERENCE, rel), os.path.join(dest, rel))
        return dest
    if os.path.exists(ARCHIVE):
        with tarfile.open(ARCHIVE, "r:gz") as tar:
            tar.extractall(dest, filter="data")
        for d, _, files in os.walk(dest):
            for f in files:
                os.chmod(os.path.join(d, f), 0o644)
        return dest
    return None


if __name__ == "__main__":
    stage(verbose=True)
