"""Build-time check of kernels that issue loads through inline asm and count `s_waitcnt vmcnt(N)` by hand.

hipcc regards the destination of an asm load as valid the moment the statement has executed, so
it is free to copy or spill that register before the hand-placed wait — which reads the register
while the load is still in flight. Nothing in the language forbids it; this lint makes the build
fail instead of producing a kernel that is wrong for some shapes. It walks the generated ISA of
every matching kernel in program order, keeps the in-order queue of outstanding `global_load`
destinations, retires all but the N youngest at each `s_waitcnt vmcnt(N)`, and reports any
instruction that names a register whose load has not been retired.

    python -m distributed_llm_inference_amd.ops.isa_lint file.s [kernel-name-regex]
"""
from __future__ import annotations

import re
import sys
from typing import List, Set, Tuple


def _regs(tok: str) -> Set[int]:
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def lint_asm(text: str, kernel_regex: str = r".") -> List[Tuple[str, str]]:
    """Returns [(kernel symbol, offending instruction)]."""
    bad: List[Tuple[str, str]] = []
    for km in re.finditer(r"^(\w+):[ \t]*(?:;.*)?$", text, re.M):
        name = km.group(1)
        if not re.search(kernel_regex, name):
            continue
        nxt = text.find(".amdhsa_kernel " + name + "\n", km.end())
        if nxt < 0:
            continue                                  # a label that is not a kernel entry
        pending: List[Set[int]] = []
        for line in text[km.end():nxt].split("\n"):
            o = line.split(";")[0].strip()
            if not o or o.startswith(".") or o.endswith(":"):
                continue
            parts = o.replace(",", " ").split()
            op, toks = parts[0], parts[1:]
            if op == "s_waitcnt":
                m = re.search(r"vmcnt\((\d+)\)", o)
                if m:
                    n = int(m.group(1))
                    pending = pending[len(pending) - n:] if 0 < n < len(pending) else ([] if n == 0 else pending)
                continue
            if op.startswith("global_load_lds"):
                # LDS-DMA: every operand is a source, nothing lands in a register; it still takes
                # a place in the in-order vmcnt queue
                src = set().union(*[_regs(t) for t in toks]) if toks else set()
                if any(p & src for p in pending):
                    bad.append((name, o))
                pending.append(set())
                continue
            if op.startswith(("global_load", "buffer_load", "flat_load")):
                src = set().union(*[_regs(t) for t in toks[1:]]) if len(toks) > 1 else set()
                if any(p & src for p in pending):
                    bad.append((name, o))
                pending.append(_regs(toks[0]))
                continue
            used = set().union(*[_regs(t) for t in toks]) if toks else set()
            if any(p & used for p in pending):
                bad.append((name, o))
    return bad


def main(argv=None) -> int:
    argv = sys.argv[1:] if argv is None else argv
    if not argv:
        print(__doc__)
        return 2
    with open(argv[0]) as f:
        bad = lint_asm(f.read(), argv[1] if len(argv) > 1 else r".")
    for k, o in bad:
        print(f"{k}: touches a register with a load in flight: {o}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
