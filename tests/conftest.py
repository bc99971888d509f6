import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# MI_PROPERTY_SOAK=<factor>: the property tests (hypothesis) leave their fixed seed and run <factor> times their
# examples -- every run of the suite explores the same cases by design (derandomize=True: a red test must stay red);
# a soak explores new ones.  tools/property_soak.sh runs it against the ASan + UBSan build of the library.
_SOAK = float(os.environ.get("MI_PROPERTY_SOAK", "0") or 0)
if _SOAK > 0:
    import hypothesis
    _init = hypothesis.settings.__init__

    def _soak_init(self, parent=None, **kw):
        if kw.get("derandomize") is True and "max_examples" in kw:      # a test's own settings, not the library's defaults
            kw["max_examples"] = max(1, int(kw["max_examples"] * _SOAK))
            kw["derandomize"] = False
        _init(self, parent, **kw)
    hypothesis.settings.__init__ = _soak_init


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import mi_oracle
    mi_oracle.build()
    return mi_oracle


@pytest.fixture(scope="session")
def engine_lib():
    import makisu_amd
    return makisu_amd.load_library()


GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GO_LAYER_TAR_DIGEST = "4ac76077f2c741c856a2419dfdb0804b18e48d2e1a9ce9c6a3f0605a2078caba"


@pytest.fixture(scope="session")
def go_layer_tar():
    """(tar bytes, member table): the gunzip of the reference's alpine layer blob (the blob's digest
    393ccd5c... is lib/utils/testutil/constants.go:28; the tar is testdata/files/busybox/393ccd5c.../
    layer.tar, written by Go's archive/tar) and tests/golden/go_layer_tar_members.json, made from that
    file with python tarfile (tests/golden/make_golden.py)."""
    import base64
    import hashlib
    import json
    import zlib
    vec = json.load(open(os.path.join(GOLDEN, "sha256_reference_fixtures.json")))["vectors"][0]
    assert vec["name"] == "alpine_layer_blob"
    raw = zlib.decompress(base64.b64decode(vec["file_b64"]), 31)
    table = json.load(open(os.path.join(GOLDEN, "go_layer_tar_members.json")))
    assert len(raw) == table["tar_bytes"] == 1308672
    assert hashlib.sha256(raw).hexdigest() == table["tar_sha256"] == GO_LAYER_TAR_DIGEST
    return raw, table["members"]
