"""Print the instruction mix of selected kernels from a hipcc -S device listing (profiling aid).
usage: hipcc -O3 --offload-arch=gfx950 <the Makefile's flags> -S --cuda-device-only csrc/bibim_hip.hip -o x.s
       python tools/isa_mix.py x.s [mangled-name fragments; default: the three per-frame kernels at 32 x 32 tiles]"""
import re, sys
from collections import Counter
path = sys.argv[1]; pats = sys.argv[2:] or ['k_geometryILi32ELi32ELb0E', 'k_rasterILi32ELi32ELb0E', 'k_shadeILi32ELi32ELb0ELb0ELb0E']
lines = open(path).read().split('\n')
cur = None; funcs = {}
for ln in lines:
    m = re.match(r'^(_ZN3bbr[A-Za-z0-9_]+):', ln)
    if m:
        cur = m.group(1); funcs[cur] = []
    elif cur is not None:
        if ln.startswith('\t.end_amdhsa_kernel') or ln.startswith('.Lfunc_end'):
            cur = None
        else:
            m2 = re.match(r'^\t([a-z][a-z_0-9]+)', ln)
            if m2: funcs[cur].append(m2.group(1))
KEYS = ['global_load_dword', 'global_load_dwordx2', 'global_load_dwordx3', 'global_load_dwordx4', 'global_load_ubyte', 'global_store_dword',
        'global_store_dwordx2', 'global_store_dwordx4', 'scratch_load_dword', 'scratch_store_dword', 'scratch_load_dwordx4', 'scratch_store_dwordx4',
        'v_mad_u64_u32', 'v_mad_i64_i32', 'v_mul_lo_u32', 'v_mul_hi_u32', 'v_mul_hi_i32', 'v_rcp_f32', 'v_sqrt_f32', 'v_rsq_f32', 'v_div_scale_f32',
        'v_div_fmas_f32', 'v_div_fixup_f32', 'v_rcp_f64', 'v_div_scale_f64', 's_load_dword', 's_load_dwordx2', 's_load_dwordx4', 's_load_dwordx8', 's_load_dwordx16',
        'ds_max_u64', 'ds_max_rtn_u64', 'global_atomic_umax_x2', 'global_atomic_umax_x2_rtn', 'global_atomic_add', 'global_atomic_add_rtn', 's_waitcnt', 'v_fma_f32', 'v_mul_f32',
        'v_add_f32', 'v_sub_f32', 'v_fmac_f32', 'v_cvt_f32_ubyte0', 'v_cvt_f32_ubyte1', 'v_cvt_f32_ubyte2', 'v_cvt_f32_ubyte3', 'v_cndmask_b32', 's_barrier', 'v_readfirstlane_b32', 's_cbranch_execz', 's_cbranch_vccnz',
        'v_pk_fma_f32', 'v_pk_mul_f32', 'v_pk_add_f32', 'v_mad_i32_i24', 'v_mad_u32_u24', 'v_mul_i32_i24', 'v_mul_u32_u24', 'ds_read_b32', 'ds_read2_b32',
        'ds_read_b128', 'ds_write_b32', 'ds_max_u64']
for name, ops in funcs.items():
    if not any(p in name for p in pats): continue
    c = Counter(ops)
    print(name[:70], 'total', len(ops))
    print('   ' + ', '.join(f'{k}={c[k]}' for k in KEYS if c.get(k)))
