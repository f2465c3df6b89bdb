"""Instruction mix of the largest loop of every kernel in a gfx950 assembly listing (tuning aid).

usage: hipcc ... -S --cuda-device-only file.hip -o file.s && python tools/loopmix.py file.s [filter]
"""
import collections
import re
import sys


def classify(op):
    table = (("v_pk_fma", "pk_fma"), ("v_pk_mul", "pk_muladd"), ("v_pk_add", "pk_muladd"), ("v_pk_mov", "pk_mov"),
             ("v_fma", "fma"), ("v_fmac", "fma"), ("v_mul_f32", "muladd"), ("v_add_f32", "muladd"),
             ("v_mov", "v_mov"), ("v_accvgpr", "acc"), ("v_", "v_other"), ("ds_read", "ds_read"), ("ds_load", "ds_read"),
             ("ds_write", "ds_write"), ("ds_store", "ds_write"), ("buffer_load", "buf_ld"), ("buffer_store", "buf_st"),
             ("global_load", "glb_ld"), ("global_store", "glb_st"), ("scratch", "scratch"), ("s_waitcnt", "waitcnt"),
             ("s_barrier", "barrier"), ("s_", "salu"))
    for pre, k in table:
        if op.startswith(pre):
            return k
    return op


def main():
    s = open(sys.argv[1]).read()
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    ks = re.split(r"\n(_Z\w+):[^\n]*\n", s)
    for i in range(1, len(ks), 2):
        name, body = ks[i], ks[i + 1]
        if flt not in name or "s_endpgm" not in body:
            continue
        lines = body[:body.index("s_endpgm")].split("\n")
        labels = {}
        for n, l in enumerate(lines):
            m = re.match(r"(\.LBB\d+_\d+):", l)
            if m:
                labels[m.group(1)] = n
        best = None
        for n, l in enumerate(lines):
            m = re.search(r"s_c?branch\w* (\.LBB\d+_\d+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] < n:
                span = (labels[m.group(1)], n)
                if best is None or span[1] - span[0] > best[1] - best[0]:
                    best = span
        if best is None:
            print(name, "no loop")
            continue
        cnt = collections.Counter()
        for l in lines[best[0]:best[1]]:
            l = l.strip()
            if not l or l.startswith((".", ";")) or l.endswith(":"):
                continue
            cnt[classify(l.split()[0])] += 1
        valu = sum(v for k, v in cnt.items() if k in ("pk_fma", "pk_muladd", "pk_mov", "fma", "muladd", "v_mov", "acc", "v_other"))
        print(name[-48:], "loop lines", best[1] - best[0], "VALU", valu, dict(cnt))


if __name__ == "__main__":
    main()
