"""TEST INFRASTRUCTURE ONLY -- never imported by the product package `ld_amd`.

Packs the reference's own Python sources for the LD train step (the `mmdet`
package and the config trees configs/{_base_,ld,gfl,ldv2}) from /root/reference
into ONE git-ignored archive, oracle/_ref/reference_snapshot.tar.gz, so that the
reference ITSELF can be timed on the GPU box's host cores beside the GPU number
(bench.py `cpu_baseline.kind == "reference"`).  The
archive is the analogue of a compiled oracle/_ref/*.so: a build product that
travels with the gpurun snapshot and never enters the git history; nothing in
`ld_amd/` reads it.  __graft_entry__.build() calls this when /root/reference is
present; on the GPU box (no /root/reference) the prebuilt archive is used as is.

    python oracle/make_ref_snapshot.py
"""
import hashlib
import io
import json
import os
import tarfile

HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.path.join(HERE, '_ref')
ARCHIVE = os.path.join(OUT_DIR, 'reference_snapshot.tar.gz')
MANIFEST = os.path.join(OUT_DIR, 'reference_snapshot.json')
TREES = ('mmdet', 'configs/_base_', 'configs/ld', 'configs/gfl', 'configs/ldv2')


def make(reference_root='/root/reference'):
    if not os.path.isdir(os.path.join(reference_root, 'mmdet')):
        return None
    os.makedirs(OUT_DIR, exist_ok=True)
    files = []
    for tree in TREES:
        for root, _, names in os.walk(os.path.join(reference_root, tree)):
            for n in sorted(names):
                if n.endswith('.py'):
                    files.append(os.path.relpath(os.path.join(root, n),
                                                 reference_root))
    files.sort()
    import gzip
    buf = io.BytesIO()
    # reproducible archive: no timestamps in the gzip header or the members
    with gzip.GzipFile(fileobj=buf, mode='wb', mtime=0) as gz:
        with tarfile.open(fileobj=gz, mode='w') as tar:
            for rel in files:
                info = tar.gettarinfo(os.path.join(reference_root, rel), arcname=rel)
                info.mtime = 0
                info.uid = info.gid = 0
                info.uname = info.gname = ''
                with open(os.path.join(reference_root, rel), 'rb') as f:
                    tar.addfile(info, f)
    data = buf.getvalue()
    with open(ARCHIVE, 'wb') as f:
        f.write(data)
    man = dict(files=len(files), bytes=len(data), trees=list(TREES),
               sha1=hashlib.sha1(data).hexdigest(), source=reference_root)
    with open(MANIFEST, 'w') as f:
        json.dump(man, f, indent=1)
    return man


def extract(dest):
    """Unpack the archive under `dest` (a scratch directory) and return it as
    the LD_REFERENCE_ROOT for oracle/ref_shim.py; None when no archive exists."""
    if not os.path.exists(ARCHIVE):
        return None
    with tarfile.open(ARCHIVE, 'r:gz') as tar:
        tar.extractall(dest)
    return dest


if __name__ == '__main__':
    print(make())
