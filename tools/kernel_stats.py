#!/usr/bin/env python3
"""Per-kernel instruction mix and register use of the exact build's trace kernels (device assembly).
usage: tools/kernel_stats.py [extra hipcc flags]   -- run from the repository root"""
import re, subprocess, sys, collections, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = "/tmp/rtamd_k_exact.s"
cmd = ["/opt/rocm/bin/hipcc", "-std=c++17", "-O3", "--offload-arch=gfx950", "-fno-slp-vectorize", "-DRT_TRACE_WAVES=4",
       "-ffp-contract=off", "-DRT_FAST=0", "-DRT_NS=rt_exact", "-S", "--cuda-device-only", "-o", out,
       os.path.join(root, "rust-raytracer_amd/csrc/rt_kernels.hip")] + sys.argv[1:]
subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
s = open(out).read()
regs = {}
for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', s, re.S):
    regs[m.group(1)] = (re.search(r'next_free_vgpr (\d+)', m.group(2)).group(1), re.search(r'next_free_sgpr (\d+)', m.group(2)).group(1),
                        re.search(r'private_segment_fixed_size (\d+)', m.group(2)).group(1))
for m in re.finditer(r'^(_ZN8rt_exact\S+):[^\n]*\n(.*?)\.Lfunc_end', s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if 'NoCounters' not in name or not ('k_shadow' in name or 'k_intersect' in name or 'k_deep' in name):
        continue
    ins = [l.split()[0] for l in body.split('\n') if l.startswith('\t') and not l.startswith('\t.') and not l.startswith('\t;')]
    c = collections.Counter(ins)
    kind = lambda p: sum(v for k, v in c.items() if k.startswith(p))
    t = re.match(r'_ZN8rt_exact\d+(k_\w+?)I(.*?)EEvN5rtamd', name)  # kernel and its template arguments, roughly
    short = t.group(1) + "<" + re.sub(r'NS_\d+', '', t.group(2).replace("Lb1", "true").replace("Lb0", "false").replace("Li", "").replace("E", ",")) + ">" if t else name
    print(f"{short:55s} instr {len(ins):6d} valu {kind('v_'):6d} salu {kind('s_') - kind('s_load'):5d} vmem {kind('global_') + kind('buffer_'):4d} "
          f"smem {kind('s_load'):4d} lds {kind('ds_'):4d} scratch {kind('scratch_'):3d}  vgpr/sgpr/private {'/'.join(regs.get(name, ('?', '?', '?')))}")
