"""Host-side native code under AddressSanitizer + UndefinedBehaviorSanitizer (sanitizers on the CPU
build only - the GPU pool offers none): csrc/pack.cpp and oracle/vad_oracle.c, built with -fsanitize=address,undefined
by tools/sanitize_host.py and driven in a child Python that has the ASan runtime preloaded."""
import importlib.util

import pytest

from conftest import REPO

spec = importlib.util.spec_from_file_location("_sanitize_host", REPO / "tools" / "sanitize_host.py")
sanitize_host = importlib.util.module_from_spec(spec)
spec.loader.exec_module(sanitize_host)


def test_packers_and_c_oracle_are_clean_under_asan_ubsan():
    try:
        sanitize_host.asan_runtime()
    except FileNotFoundError as e:                       # a toolchain without compiler-rt: nothing to run the check with
        pytest.skip(str(e))
    r = sanitize_host.run()
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-6000:])
    assert "no ASan / UBSan report" in r.stdout
