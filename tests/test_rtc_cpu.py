"""Run-time compilation plumbing that needs no device (libhiprtc compiles without one): the user-macro scanner and the
on-disk code-object cache of csrc/mm_rtc.hip, through the developer library's compile hook."""

import os
import subprocess
import sys

import pytest

from conftest import ROOT

DEV = os.path.join(ROOT, "mici_amd", "lib", "libmici_amd_dev.so")
pytestmark = pytest.mark.skipif(not os.path.exists(DEV), reason="developer library not built")

sys.path.insert(0, os.path.join(ROOT, "tests"))

_SNIPPET = r"""
import ctypes as C, sys
lib = C.CDLL(sys.argv[1])
lib.mm_debug_rtc_compile.restype = C.c_long
lib.mm_debug_rtc_compile.argtypes = [C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_char_p]
lib.mm_last_error.restype = C.c_char_p
lib.mm_last_error.argtypes = [C.c_void_p]
src = open(sys.argv[2]).read()
n = lib.mm_debug_rtc_compile(int(sys.argv[3]), 0, 0, src.encode(), None)
print(n)
print(lib.mm_last_error(None).decode("utf-8", "replace")[:1500] if n < 0 else "")
"""


def _compile(tmp_path, text, dim=4, cache=None):
    """mm_debug_rtc_compile(dim, built-in target, wave family) in a FRESH interpreter (the in-process code cache starts
    empty); returns (size or negative rc, error text)."""
    src = tmp_path / "src.hip"
    src.write_text(text)
    env = dict(os.environ, MICI_AMD_RTC_SEED="off", MICI_AMD_RTC_CACHE=str(cache) if cache else "off")
    out = subprocess.run([sys.executable, "-c", _SNIPPET, DEV, str(src), str(dim)], env=env, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.splitlines()
    return int(lines[0]), "\n".join(lines[1:])


def test_user_aux_must_be_a_plain_integer(tmp_path):
    from user_sources import RANK1_AS_USER
    n, msg = _compile(tmp_path, "#define MM_USER_AUX (2 * 64 + 2)\n" + RANK1_AS_USER)
    assert n < 0 and "plain decimal integer" in msg, (n, msg)
    n, msg = _compile(tmp_path, "#define MM_USER_AUX 0\n" + RANK1_AS_USER)
    assert n < 0 and "1 .. 560" in msg, (n, msg)
    n, msg = _compile(tmp_path, "#define MM_USER_AUX 561\n" + RANK1_AS_USER)
    assert n < 0 and "1 .. 560" in msg, (n, msg)


def test_opt_in_macros_inside_comments_and_strings_are_not_hoisted(tmp_path):
    """A commented-out `#define MM_USER_VJP_FLAT` used to switch the headers to the team-form VJP hook, which the text then
    did not define (an undefined-symbol error far from the cause).  The plain source with such comments must compile as
    the plain source."""
    from user_sources import RANK1_AS_USER
    text = ("// #define MM_USER_VJP_FLAT\n/* disabled:\n#define MM_USER_AUX 130\n#define MM_USER_VJP_FLAT 1\n*/\n"
            "static __device__ const char* kNote = \"#define MM_USER_VJP_FLAT\";\n" + RANK1_AS_USER)
    n, msg = _compile(tmp_path, text)
    assert n > 0, msg


def test_damaged_cache_file_is_recompiled_not_loaded(tmp_path):
    """A truncated / foreign .hsaco in the cache directory must not be handed to the runtime: it is not an ELF image, the
    text is compiled again and the file replaced (atomic rename of a mkstemp file)."""
    from user_sources import RANK1_AS_USER
    cache = tmp_path / "cache"
    n1, msg = _compile(tmp_path, RANK1_AS_USER, cache=cache)
    assert n1 > 0, msg
    files = [f for f in os.listdir(cache) if f.endswith(".hsaco")]
    assert len(files) == 1 and not [f for f in os.listdir(cache) if f.startswith(".tmp_")]
    path = cache / files[0]
    good = path.read_bytes()
    assert good[:4] == b"\x7fELF" and len(good) == n1
    path.write_bytes(good[:100].replace(b"\x7fELF", b"JUNK"))  # truncated AND without the magic
    n2, msg = _compile(tmp_path, RANK1_AS_USER, cache=cache)
    assert n2 == n1, (n2, msg)
    assert path.read_bytes() == good  # recompiled (deterministically) and stored again
    # truncated only - the magic is there, the section table the header points at is not: handed to the runtime such an
    # image aborts the process inside hipModuleLoadData (seen on the GPU box), so the bounds check must reject it
    for cut in (2000, len(good) // 2, len(good) - 1):
        path.write_bytes(good[:cut])
        n3, msg = _compile(tmp_path, RANK1_AS_USER, cache=cache)
        assert n3 == n1, (cut, n3, msg)
        assert path.read_bytes() == good
