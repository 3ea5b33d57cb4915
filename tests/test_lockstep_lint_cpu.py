"""CPU: the meeting-point invariant of the lock-step kernels, enforced textually (ccqp_lockstep.h).

A device function that contains a meeting point (`bp_meet(`) -- or that receives the entry snapshot of one (`*_words& z`)
-- may have its problem's slab changed by the block that arrives last while other blocks of the same launch are still
starting.  Such a function therefore never READS the slab: it takes the by-value snapshot `*_enter(slab)` (which ends
in a workgroup barrier) first and writes the slab only, computing from the snapshot.  The rule checked here:

  in every function whose body mentions `bp_meet(` or `*_enter(`, or whose signature takes a `_words&` snapshot, each textual use of
  the slab (`q.slab[...]`, `slab[...]`, `s[...]` where `s` aliases the slab) is the target of a plain assignment, and the
  slab pointer is handed to nothing but an `_enter(` helper, `bbf_decide_a(`, `bapgd_decide_scalars<` or `mp_tail<`
  (themselves checked).

A regression of round 1's race (a late wave reading the NEW state once per ~10^4 decisions) would need someone to add
such a read; this test is what stops it."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ccqppy_amd", "csrc")
FILES = ["ccqp_lockstep.h", "ccqp_bpipe.hip", "ccqp_bmprgp.hip", "ccqp_bsegs.hip", "ccqp_bspg.hip", "ccqp_pipe.hip"]


def _strip_comments(src):
    src = re.sub(r"/\*.*?\*/", lambda m: " " * len(m.group(0)), src, flags=re.S)
    return re.sub(r"//[^\n]*", "", src)


def _functions(src):
    """(name, signature, body) of every function definition: a `)` followed by `{` at nesting depth 0."""
    out, depth, i, start_sig = [], 0, 0, 0
    n = len(src)
    while i < n:
        c = src[i]
        if c == "{":
            if depth == 0:
                # signature = text since the previous top-level `}` or `;`
                sig = src[start_sig:i]
                j, d = i, 0
                while j < n:
                    if src[j] == "{":
                        d += 1
                    elif src[j] == "}":
                        d -= 1
                        if d == 0:
                            break
                    j += 1
                body = src[i:j + 1]
                m = re.findall(r"(\w+)\s*\(", sig)
                if m and ")" in sig and "struct" not in sig.split("(")[0].split()[-2:-1]:
                    out.append((m[0] if m[0] not in ("__launch_bounds__",) else (m[1] if len(m) > 1 else m[0]), sig, body))
                i = j
                start_sig = j + 1
            depth = 0
        elif c == ";" and depth == 0:
            start_sig = i + 1
        i += 1
    return out


SLAB_USE = re.compile(r"(?<![\w.])(?:q\.slab|slab|s)\s*\[")


def _slab_reads(body):
    """Textual uses of the slab in `body` that are not plain assignment targets."""
    bad = []
    for m in SLAB_USE.finditer(body):
        # find the matching `]`
        j, d = m.end(), 1
        while d and j < len(body):
            d += body[j] == "["
            d -= body[j] == "]"
            j += 1
        rest = body[j:j + 4].lstrip()
        if not (rest.startswith("=") and not rest.startswith("==")):
            line = body[body.rfind("\n", 0, m.start()) + 1: body.find("\n", m.start())].strip()
            bad.append(line)
    return bad


def _guarded(name, sig, body):
    return "bp_meet(" in body or re.search(r"\b(bp|ap|mp)_enter\s*\(", body) is not None \
        or re.search(r"_words\s*&", sig) is not None


@pytest.mark.parametrize("fname", FILES)
def test_no_slab_read_in_a_function_with_a_meeting_point(fname):
    src = _strip_comments(open(os.path.join(CSRC, fname)).read())
    checked = 0
    for name, sig, body in _functions(src):
        if name.endswith("_enter") or name == "bp_meet":
            continue
        if not _guarded(name, sig, body):
            continue
        checked += 1
        # `s` must alias the slab only (`double* s = q.slab;`) if it is used with [] at all
        if re.search(r"(?<![\w.])s\s*\[", body):
            assert re.search(r"double\s*\*\s*s\s*=\s*q\.slab\s*;", body) or re.search(r"double\s*\*\s*s\b", sig), \
                "%s:%s: `s[...]` is used but `s` is not declared as the slab alias" % (fname, name)
        bad = _slab_reads(body)
        assert not bad, "%s:%s reads the slab after its entry snapshot:\n  %s" % (fname, name, "\n  ".join(bad))
        # the slab pointer itself travels only to the snapshot helpers and the checked decision helpers
        for m in re.finditer(r"(?<![\w.])(?:q\.slab|slab|s)\b(?!\s*[\[\w=])", body):
            if m.group(0) == "s" and not re.search(r"double\s*\*\s*s\b", sig + body):
                continue
            before = body[:m.start()]
            if re.search(r"double\s*\*\s*s\s*=\s*$", before) or re.search(r"double\s*\*\s*$", before):
                continue                              # the alias declaration itself
            # the innermost call this use is an argument of
            d, j = 0, m.start() - 1
            while j >= 0:
                if body[j] == ")":
                    d += 1
                elif body[j] == "(":
                    if d == 0:
                        break
                    d -= 1
                j -= 1
            callee = re.search(r"(\w+)\s*(?:<[^<>]*>)?\s*$", body[:j])
            callee = callee.group(1) if callee else "?"
            assert callee.endswith("_enter") or callee in ("bbf_decide_a", "mp_tail", "bapgd_decide_scalars"), \
                "%s:%s hands the slab to %s()" % (fname, name, callee)
    if fname in ("ccqp_bpipe.hip", "ccqp_bmprgp.hip", "ccqp_bsegs.hip"):
        assert checked >= 3, "the lint found no function with a meeting point in %s -- the parser is broken" % fname


def test_every_function_with_a_meeting_point_takes_the_entry_snapshot():
    """... unless it reads nothing of the slab at all (start-up kernels that only write it) -- or IS handed the snapshot."""
    for fname in FILES:
        src = _strip_comments(open(os.path.join(CSRC, fname)).read())
        for name, sig, body in _functions(src):
            if name == "bp_meet" or "bp_meet(" not in body:
                continue
            takes = re.search(r"\b(bp|ap|mp)_enter\s*\(", body) or re.search(r"_words\s*&", sig) \
                or re.search(r"const double \w+0 = slots\[", body)
            uses_z = re.search(r"\bz\.\w+", body)
            if uses_z:
                assert takes, "%s:%s uses a snapshot it never took" % (fname, name)


def test_the_lint_catches_a_planted_read():
    planted = """
    __global__ void k(const ccqp_pprob* probs) {
      const ccqp_pprob& q = probs[blockIdx.y];
      const bp_words z = bp_enter(q.slab);
      double v[1];
      if (!bp_meet(q.partials, q.counter, 1, 1, v, 0, 0)) return;
      if (threadIdx.x == 0) q.slab[BP_ITERS] = q.slab[BP_ITERS] + 1.0;
    }
    """
    fns = _functions(_strip_comments(planted))
    assert len(fns) == 1 and fns[0][0] == "k"
    assert _slab_reads(fns[0][2]) == ["if (threadIdx.x == 0) q.slab[BP_ITERS] = q.slab[BP_ITERS] + 1.0;"]
