"""TEST INFRASTRUCTURE: pack the reference's Python tree for the GPU box.

/root/reference does not exist on the MI355X box, but tests/test_gpu_reference_unchanged.py has to run the reference's OWN,
unchanged lib/net + lib/rpn + lib/utils Python on the HIP kernels there.  Like oracle/_ref/libprcnn_ref.so (the reference's
native sources compiled for the host), this is a BUILT ARTEFACT of the checker: one archive under oracle/_ref/ (git-ignored,
never committed, not gpurun-ignored so that it travels with the snapshot).  Nothing in the product reads it; the test unpacks
it into a temporary directory and sets sys.path exactly as tools/_init_path.py:1-4 does.

    python -m oracle.stage_reference            # -> oracle/_ref/reference_py.tar.gz
"""
import io
import os
import tarfile

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE = os.environ.get("PRCNN_REFERENCE", "/root/reference")
ARCHIVE = os.path.join(HERE, "_ref", "reference_py.tar.gz")
# what the network-level route needs: the library package, the config files and the path helper of tools/
WANTED = (("lib", (".py",)), ("tools/cfgs", (".yaml",)), ("tools", ("_init_path.py",)))


def have_reference():
    return os.path.isdir(os.path.join(REFERENCE, "lib", "net"))


def members():
    out = []
    for sub, suffixes in WANTED:
        top = os.path.join(REFERENCE, sub)
        for d, dirs, files in os.walk(top):
            dirs[:] = sorted(x for x in dirs if x != "__pycache__" and (sub != "tools" or d != top))
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


def locate(tmpdir=None):
    """-> a directory holding the reference tree (lib/, tools/cfgs/): $PRCNN_REFERENCE or /root/reference when present, else the
    staged archive unpacked into `tmpdir`; None when neither exists"""
    if have_reference():
        return REFERENCE
    if os.path.exists(ARCHIVE) and tmpdir is not None:
        with tarfile.open(ARCHIVE, "r:gz") as tar:
            tar.extractall(tmpdir)
        return str(tmpdir)
    return None


if __name__ == "__main__":
    stage(verbose=True)
