"""CPU: two textual rules over ccqppy_amd/csrc/ (round 4, VERDICT r03 item 6 / ADVICE r03).

1. The product library never reads the ENVIRONMENT: which kernel a caller's solve runs as is decided by the arguments
   (the `variant` / `flags` words) and by per-context options (`ccqp_ctx_set_option`, include/ccqp_hip.h) alone.  A
   `getenv(` in a launch path changes measurements silently, is invisible in the C ABI and races with `setenv`.
2. No process-wide once-flag guards `hipFuncSetAttribute(MaxDynamicSharedMemorySize)`: HIP keeps that attribute per
   DEVICE, so the record of "already raised" lives in the context (`ccqp_lds_raise`, ccqp_ctx.hip) -- a function-local
   `static bool raised` would leave a second device of the same process with a > 64 KiB launch that fails.
   `hipFuncSetAttribute` itself is called from exactly one place."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ccqppy_amd", "csrc")


def _sources():
    for name in sorted(os.listdir(CSRC)):
        if name.endswith((".hip", ".h")):
            with open(os.path.join(CSRC, name)) as fh:
                src = fh.read()
            src = re.sub(r"/\*.*?\*/", lambda m: re.sub(r"[^\n]", " ", m.group(0)), src, flags=re.S)
            yield name, re.sub(r"//[^\n]*", "", src)


def _hits(pattern):
    out = []
    for name, src in _sources():
        for m in re.finditer(pattern, src):
            out.append("%s:%d: %s" % (name, src.count("\n", 0, m.start()) + 1, m.group(0)))
    return out


def test_no_environment_reads_in_the_library():
    assert _hits(r"\b(?:secure_)?getenv\s*\(|\benviron\b") == []


def test_no_function_local_once_flags_for_the_lds_attribute():
    assert _hits(r"\bstatic\s+(?:const\s+)?(?:bool|int|size_t|unsigned)\s+raised\b") == []
    calls = _hits(r"\bhipFuncSetAttribute\s*\(")
    assert len(calls) == 1 and calls[0].startswith("ccqp_ctx.hip:"), calls


def test_every_option_of_the_header_has_a_name_in_the_binding():
    from ccqppy_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "ccqp_hip.h")).read()
    keys = dict(re.findall(r"CCQP_OPT_(\w+)\s*=\s*(\d+)", hdr))
    count = int(keys.pop("COUNT"))
    assert {k: int(v) for k, v in keys.items()} == _lib.OPTIONS and count == len(_lib.OPTIONS)
    assert int(re.search(r"#define CCQP_ABI_VERSION (\d+)", hdr).group(1)) == _lib.ABI_VERSION
    assert int(re.search(r"#define CCQP_LOCKSTEP_LOOKAHEAD \(1 << (\d+)\)", hdr).group(1)) == 17 and _lib.LOCKSTEP_LOOKAHEAD == 1 << 17
