"""GPU test of the layer-tar staging path (mi_batch_add_path_range / Batch.add_tar): the regular
files of a layer tar scanned straight out of the archive."""
import hashlib

import numpy as np
import pytest

try:
    import torch  # noqa: F401  (same load-order rule as tests/test_gpu_parity.py)
except ImportError:
    torch = None

pytestmark = pytest.mark.gpu

SEED = 0x4D414B49


def test_layer_tar_members_scanned_in_place(oracle, tmp_path):
    """mi_tar_entries + mi_batch_add_path_range: the regular files of a layer tar hashed straight out
    of the archive equal the same bytes added one by one (whole-file SHA-256 vs hashlib, chunk
    roots vs the oracle)."""
    import io
    import tarfile
    import makisu_amd
    blobs = {"bin/tool": oracle.synth_fill(SEED, 900, 0, 300000).tobytes(), "etc/empty": b"",
             "etc/one": b"x", "lib/block": oracle.synth_fill(SEED, 901, 0, 65536).tobytes()}
    p = str(tmp_path / "layer.tar")
    with tarfile.open(p, "w", format=tarfile.GNU_FORMAT) as tf:
        for d in ("bin/", "etc/", "lib/"):
            ti = tarfile.TarInfo(d)
            ti.type = tarfile.DIRTYPE
            tf.addfile(ti)
        for name, data in blobs.items():
            ti = tarfile.TarInfo(name)
            ti.size = len(data)
            tf.addfile(ti, io.BytesIO(data))
    with makisu_amd.Engine(flags=makisu_amd.FLAG_FILE_SHA256) as e, e.batch() as b:
        ents = b.add_tar(p)
        b.run()
        files = b.files().copy()
    regs = [x for x in ents if x["kind"] == 1]
    assert [x["relpath"] for x in regs] == list(blobs) and [x["file_index"] for x in regs] == [0, 1, 2, 3]
    datas = list(blobs.values())
    for row, data in zip(files, datas):
        assert int(row["size"]) == len(data)
        assert row["file_sha256"].tobytes() == hashlib.sha256(data).digest()
    arr = np.frombuffer(b"".join(datas), dtype=np.uint8)
    sizes = np.array([len(d) for d in datas], dtype=np.uint64)
    offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.uint64)
    rf, _ = oracle.scan_batch(arr, offs, sizes, oracle.CdcParams(SEED, 13, 2048, 65536))
    assert np.array_equal(files["chunk_root"], rf["chunk_root"])
