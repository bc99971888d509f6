"""GPU tests that mirror uber/makisu's OWN tests for this path, case by case, with the engine in
the place of the Go code (each test names the reference test it follows)."""
import base64
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
EMPTY_TAR_LINUX = "84ff92691f909a05b224e1c56abb4864f01b4f8e3c854e4bb4c7baf1d3f6d652"   # const_linux.go:18


def _empty_gnu_tar(tmp_path):
    """`tar cvf <target> --files-from /dev/null` as the reference tests do; 10240 zero bytes."""
    target = tmp_path / "empty.tar"
    if shutil.which("tar"):
        subprocess.check_call(["tar", "cvf", str(target), "--files-from", "/dev/null"])
    else:
        target.write_bytes(bytes(10240))
    return target


def test_digest_from_bytes(tmp_path):
    """lib/docker/image/digester_test.go:28-57 TestDigestFromBytes: FromReader(file) == FromBytes(bytes)."""
    import makisu_amd
    target = _empty_gnu_tar(tmp_path)
    with makisu_amd.Engine(flags=makisu_amd.FLAG_FILE_SHA256) as eng:
        with eng.batch() as b:                                   # "reader": the engine reads the path
            b.add_path(str(target))
            b.run()
            d1 = makisu_amd.Digest.from_raw(b.files()["file_sha256"][0])
        d2 = makisu_amd.Digest.from_raw(np.frombuffer(eng.sha256_many([target.read_bytes()])[0], np.uint8))
    assert d1 == d2
    assert d1.startswith("sha256:") and len(d1.hex()) == 64


def test_empty_digest(tmp_path):
    """lib/docker/image/digest_test.go:37-62 TestEmptyDigest: an empty GNU tar hashes to DigestEmptyTar."""
    import makisu_amd
    target = _empty_gnu_tar(tmp_path)
    with makisu_amd.Engine() as eng:
        got = eng.sha256_many([target.read_bytes()])[0]
    assert makisu_amd.Digest.from_raw(np.frombuffer(got, np.uint8)) == "sha256:" + EMPTY_TAR_LINUX


def test_layer_digest_equals_sample_layer():
    """lib/registry/client_test.go:45-62 via saveLayer (client.go:616-633) and
    lib/docker/image/digest.go:42-50 Digest.Equals: the pulled alpine layer blob verifies against
    testutil.SampleLayerTarDigest (lib/utils/testutil/constants.go:28)."""
    import makisu_amd
    gold = json.load(open(os.path.join(HERE, "golden", "sha256_reference_fixtures.json")))["vectors"]
    layer = next(v for v in gold if v["name"] == "alpine_layer_blob")
    with makisu_amd.Engine() as eng:
        computed = makisu_amd.Digest.from_raw(np.frombuffer(
            eng.sha256_many([base64.b64decode(layer["file_b64"])])[0], np.uint8))
    assert computed == "sha256:393ccd5c4dd90344c9d725125e13f636ce0087c62f5ca89050faaacbb9e3ed5b"
    assert computed.hex() == layer["sha256"]                     # Digest.Hex()


def _cache_id(eng, context_dir, seed, args, from_stage=False):
    """addCopyStep.SetCacheID (add_copy_step.go:102-122): crc32(seed+directive+args), then --
    unless the step copies from another stage -- the context walk."""
    with eng.batch() as b:
        if not from_stage:
            b.add_tree(str(context_dir), rel_base=str(context_dir))
        b.run()
        return b.context_checksum_tree((seed + "COPY" + args).encode())


@pytest.fixture()
def copy_context(tmp_path):
    """context.BuildContextFixture + the source tree the reference's subtests create
    (copy_step_test.go:53-66): ctx/<sourceDir>/<subDir>/<file with 1 KiB of random bytes>."""
    ctx = tmp_path / "context"
    sub = ctx / "testCopyStepSource123" / "testCopyStepSub456"
    os.makedirs(sub)
    f = sub / "testCopyStepFile789"
    f.write_bytes(np.random.default_rng().integers(0, 256, 1024, dtype=np.uint8).tobytes())
    return ctx, f


def test_copy_step_set_cache_id(copy_context):
    """lib/builder/step/copy_step_test.go:51-169 TestCopyStepSetCacheID, all four subtests."""
    import makisu_amd
    ctx, src_file = copy_context
    with makisu_amd.Engine(flags=makisu_amd.FLAG_FILE_CRC32) as eng:
        # CopyFromSameContext: same inputs twice -> same cache ID
        hash1 = _cache_id(eng, ctx, "", ". tmp")
        assert _cache_id(eng, ctx, "", ". tmp") == hash1
        # CopyFromSameContextDifferentSeed: seeding with the previous ID changes it
        assert _cache_id(eng, ctx, hash1, ". tmp") != hash1
        # CopyFromDifferentContexts: another destination -> different; changed content -> different
        assert _cache_id(eng, ctx, hash1, ". tmp2") != hash1
        src_file.write_bytes(b"new content")
        assert _cache_id(eng, ctx, "", ". tmp") != hash1
        # CopyFromStage: no context hashing, the ID depends on seed/args only and is stable
        s1 = _cache_id(eng, ctx, "seed", ". tmp", from_stage=True)
        assert _cache_id(eng, ctx, "seed", ". tmp", from_stage=True) == s1
        assert all(c in "0123456789abcdef" for c in s1) and len(s1) <= 8     # "%x", unpadded
