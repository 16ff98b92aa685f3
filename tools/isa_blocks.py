"""Per-basic-block instruction census of one kernel in a hipcc -S listing (development aid).
    hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=fast -fno-slp-vectorize -Iraytracer_amd/csrc -Iinclude --cuda-device-only -S raytracer_amd/csrc/kernels_small.hip -o /tmp/ks.s
    python tools/isa_blocks.py /tmp/ks.s k_pathtrace_smallILi0ELb0ELb0E [--dump LABEL]"""
import collections, re, sys
path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if key in l and l.rstrip().endswith(":") is False and re.match(r"^_Z\S+:", l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
blocks, cur = collections.OrderedDict(), "entry"
blocks[cur] = []
for l in lines[start + 1:end]:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        cur = m.group(1); blocks[cur] = []; continue
    t = l.strip()
    if t and not t.startswith((";", ".")):
        blocks[cur].append(t)
if "--dump" in sys.argv:
    want = sys.argv[sys.argv.index("--dump") + 1:]
    for b in want:
        print(b + ":"); print("\n".join("    " + t for t in blocks[b]))
    sys.exit(0)
def kind(t):
    op = t.split()[0]
    if op.startswith("v_"):
        if op.startswith(("v_rcp", "v_sqrt", "v_rsq", "v_sin", "v_cos", "v_exp", "v_log")): return "trans"
        if op.startswith(("v_mad_u64", "v_mul_lo", "v_mul_hi")): return "imul"
        if op.startswith("v_mov"): return "mov"
        if op.startswith("v_cndmask"): return "cnd"
        if op.startswith("v_cmp"): return "cmp"
        if op.startswith("v_pk"): return "pk"
        return "valu"
    if op.startswith(("s_load", "s_buffer")): return "smem"
    if op.startswith(("s_cbranch", "s_branch")): return "br"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    return "mem"
tot = collections.Counter()
for b, ins in blocks.items():
    c = collections.Counter(kind(t) for t in ins)
    tot.update(c)
    v = sum(c[k] for k in ("valu", "trans", "imul", "mov", "cnd", "cmp", "pk"))
    if ins:
        br = [t for t in ins if t.startswith(("s_cbranch", "s_branch"))]
        print(f"{b:12s} n={len(ins):4d} VALU={v:4d} (mov {c['mov']:3d} cnd {c['cnd']:3d} cmp {c['cmp']:3d} trans {c['trans']:2d} imul {c['imul']:2d} pk {c['pk']:2d}) salu {c['salu']:3d} smem {c['smem']:2d} lds {c['lds']:2d} wait {c['wait']:2d}  -> {' '.join(x.split()[-1] for x in br)}")
print(dict(tot))
