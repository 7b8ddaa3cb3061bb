"""Timing-only variants of the transform pass kernel (results are wrong on purpose): which parts of the kernel's time
are the field multiplications (M), the LDS exchange (L), the tile-twiddle loads (T) and the workgroup barriers (B).

  python tools/ablate_pass_kernel.py MLT      # writes csrc/_t.hip from csrc/ntt29.hip with those parts removed

Build the variant into its own library (see tools/ab.sh) and time it with tools/bench_one.py; never run tests on it."""
import os
import sys
os.chdir(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "stark_verifier_amd", "csrc"))
v=sys.argv[1]
s=open('ntt29.hip').read()
def fake_muls(s):
    s=s.replace('x1 = f29_mul(x1, tw1);','x1 = f29_carry(f29_add(x1, tw1));')
    s=s.replace('x2 = f29_mul(x2, tw2);','x2 = f29_carry(f29_add(x2, tw2));')
    s=s.replace('x3 = f29_mul(x3, tw3);','x3 = f29_carry(f29_add(x3, tw3));')
    s=s.replace('F29 a3 = f29_mul_uniform2(f29_sub(x1, x3), a.iq, a.iq2);','F29 a3 = f29_carry(f29_sub(x1, x3));')
    for k in range(4):
        s=s.replace('y%d = f29_mul(y%d, f29_from_fe(t%d));'%(k,k,k),'y%d = f29_carry(f29_add(y%d, f29_from_fe(t%d)));'%(k,k,k))
    return s
if 'M' in v: s=fake_muls(s)
if 'L' in v:  # no LDS traffic: puts dropped, gets return a register-derived value
    s=s.replace('''    uint32_t* q = lds + off * 9;
#pragma unroll
    for (int l = 0; l < 9; l++) q[l] = v.l[l];''','''    if (off == 0xFFFFFFFFu) lds[0] = v.l[0];''')
    s=s.replace('''    const uint32_t* q = lds + off * 9;
    F29 v;
#pragma unroll
    for (int l = 0; l < 9; l++) v.l[l] = q[l];
    return v;''','''    F29 v;
#pragma unroll
    for (int l = 0; l < 9; l++) v.l[l] = off * (l + 1);
    return v;''')
if 'T' in v:  # no tile-twiddle loads
    s=s.replace('if (ns_log > 0) tw1 = f29_load_raw(a.W + 12 * (size_t)e);','if (ns_log > 0) { for (int l = 0; l < 9; l++) tw1.l[l] = e + l; }')
    s=s.replace('F29 tw2 = f29_load_raw(a.W + 12 * (size_t)(2 * e));','F29 tw2; for (int l = 0; l < 9; l++) tw2.l[l] = 2 * e + l;')
    s=s.replace('F29 tw3 = f29_load_raw(a.W + 12 * (size_t)(3 * e));','F29 tw3; for (int l = 0; l < 9; l++) tw3.l[l] = 3 * e + l;')
if 'B' in v:  # no workgroup barriers
    s=s.replace('__syncthreads();','__builtin_amdgcn_wave_barrier();')
open('_t.hip','w').write(s)
