"""Build-time check of kernels that issue loads through inline asm and count `s_waitcnt vmcnt(N)` by hand.

hipcc regards the destination of an asm load as valid the moment the statement has executed, so
it is free to copy or spill that register before the hand-placed wait — which reads the register
while the load is still in flight. Nothing in the language forbids it; this lint makes the build
fail instead of producing a kernel that is wrong for some shapes. It walks the control-flow graph
of every matching kernel's ISA, keeps the in-order queue of outstanding `global_load`
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


def _kernel_bodies(text: str, kernel_regex: str):
    for km in re.finditer(r"^(\w+):[ \t]*(?:;.*)?$", text, re.M):
        name = km.group(1)
        if not re.search(kernel_regex, name):
            continue
        end = text.find(".amdhsa_kernel " + name + "\n", km.end())
        if end < 0:
            continue                                  # a label that is not a kernel entry
        yield name, text[km.end():end]


def lint_asm(text: str, kernel_regex: str = r".") -> List[Tuple[str, str]]:
    """Returns [(kernel symbol, offending instruction)].

    The walk follows the control-flow graph (hipcc rotates loops, so program order is not execution
    order): every instruction is checked once, under the queue state of the first path that reaches
    it; both arms of a conditional branch are explored with the state at the branch."""
    bad: List[Tuple[str, str]] = []
    for name, body in _kernel_bodies(text, kernel_regex):
        ops: List[str] = []
        labels = {}
        for line in body.split("\n"):
            o = line.split(";")[0].strip()
            if not o or o.startswith("."):
                if o.startswith(".L") and o.endswith(":"):
                    labels[o[:-1]] = len(ops)
                continue
            if o.endswith(":"):
                labels[o[:-1]] = len(ops)
                continue
            ops.append(o)
        seen = [False] * len(ops)
        stack = [(0, ())]                              # (pc, pending queue as a tuple of frozensets)
        while stack:
            pc, pend_t = stack.pop()
            pending = list(pend_t)
            while 0 <= pc < len(ops) and not seen[pc]:
                seen[pc] = True
                o = ops[pc]
                parts = o.replace(",", " ").split()
                op, toks = parts[0], parts[1:]
                nxt = pc + 1
                if op == "s_waitcnt":
                    m = re.search(r"vmcnt\((\d+)\)", o)
                    if m:
                        n = int(m.group(1))
                        pending = pending[len(pending) - n:] if 0 < n < len(pending) else ([] if n == 0 else pending)
                elif op == "s_endpgm":
                    break
                elif op == "s_branch":
                    nxt = labels.get(toks[0], len(ops))
                elif op.startswith("s_cbranch"):
                    tgt = labels.get(toks[0])
                    if tgt is not None and not seen[tgt]:
                        stack.append((tgt, tuple(pending)))
                elif op.startswith("global_load_lds"):
                    # LDS-DMA: every operand is a source, nothing lands in a register; it still takes
                    # a place in the in-order vmcnt queue
                    src = set().union(*[_regs(t) for t in toks]) if toks else set()
                    if any(p & src for p in pending):
                        bad.append((name, o))
                    pending.append(frozenset())
                elif op.startswith(("global_load", "buffer_load", "flat_load")):
                    src = set().union(*[_regs(t) for t in toks[1:]]) if len(toks) > 1 else set()
                    if any(p & src for p in pending):
                        bad.append((name, o))
                    pending.append(frozenset(_regs(toks[0])))
                else:
                    used = set().union(*[_regs(t) for t in toks]) if toks else set()
                    if any(p & used for p in pending):
                        bad.append((name, o))
                pc = nxt
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
