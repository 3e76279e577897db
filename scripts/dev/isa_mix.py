"""Static instruction mix of a kernel's hottest loop from the compiler's assembly (development; runs without a GPU).
    python scripts/dev/isa_mix.py prob3_events '_ZN4pisa19prob3_events_kernelILb0ELi0ELb0ELi3ELb0EEEv' [out.json]
Compiles pisa_amd/csrc/<file>.hip with the Makefile's flags to assembly, takes the named kernel, finds its outermost
depth-1 loop with the most instructions (the layer walk of the event kernel) and counts the instructions by class."""
import collections, json, os, re, subprocess, sys, tempfile

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src, sym = sys.argv[1], sys.argv[2]
flags = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-fno-fast-math", "-I" + root + "/include", "-I" + root + "/pisa_amd/csrc"]
flags.append("-ffp-contract=fast" if src in ("prob3_events", "prob3_planned") else "-ffp-contract=off")
with tempfile.TemporaryDirectory() as tmp:
    out = os.path.join(tmp, "k.s")
    subprocess.run(["/opt/rocm/bin/hipcc"] + flags + ["-S", "--cuda-device-only", "-o", out, "%s/pisa_amd/csrc/%s.hip" % (root, src)],
                   check=True, stderr=subprocess.DEVNULL)
    lines = open(out).read().splitlines()
start = next(i for i, l in enumerate(lines) if l.startswith(sym) and l.rstrip().endswith(":") or (l.startswith(sym) and ": " in l and "@" in l))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = lines[start:end]
heads = [i for i, l in enumerate(body) if "Loop Header: Depth=1" in l]
best = None
for h in heads:
    j = h + 1
    while j < len(body) and not (re.match(r"^\.LBB[0-9_]+:\s*$", body[j]) or ("; %bb." in body[j] and "in Loop" not in body[j])):
        j += 1
    if best is None or j - h > best[1] - best[0]:
        best = (h, j)
loop = [l.split()[0] for l in body[best[0]:best[1]] if re.match(r"^\s+(v_|s_|ds_|global_|buffer_|flat_)", l)]
def cls(op):
    if re.match(r"v_(fma|fmac)_f64", op): return "fp64 fused multiply-add"
    if re.match(r"v_mul_f64", op): return "fp64 multiply"
    if re.match(r"v_add_f64", op): return "fp64 add"
    if re.match(r"v_(rcp|rsq|sqrt|ldexp|frexp|max|min|cmp.*)_?f64|v_(max|min)_f64|v_cmp_.*f64", op): return "fp64 other (rcp / rsq / ldexp / max / compare)"
    if op.startswith("v_readlane") or op.startswith("v_writelane"): return "scalar-register spill traffic (v_readlane / v_writelane)"
    if op.startswith("v_cndmask"): return "select (v_cndmask)"
    if op.startswith("v_mov"): return "register copy (v_mov)"
    if op.startswith("v_"): return "integer / address / other vector"
    if op.startswith("ds_"): return "LDS (not a vector-ALU instruction)"
    if op.startswith("s_"): return "scalar (not a vector-ALU instruction)"
    return "memory"
cnt = collections.Counter(cls(op) for op in loop)
valu = sum(v for k, v in cnt.items() if "not a vector" not in k and k != "memory")
res = {"kernel": sym, "loop_instructions": len(loop), "vector_alu_instructions": valu,
       "by_class": dict(cnt.most_common()), "fraction_of_vector_alu": {k: round(v / valu, 4) for k, v in cnt.most_common() if "not a vector" not in k and k != "memory"},
       "method": "static count over the compiler's assembly of the kernel's largest depth-1 loop (scripts/dev/isa_mix.py); the loop holds both inlined copies of the layer amplitude"}
print(json.dumps(res, indent=1))
if len(sys.argv) > 3:
    json.dump(res, open(sys.argv[3], "w"), indent=1)
