#!/usr/bin/env python3
"""Per-kernel ISA summary (VGPRs, scratch, MFMA / accvgpr / LDS / s_nop counts) of a hipcc --save-temps .s file.
   usage: isa_stats.py file.s [name-filter]"""
import re, sys
s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r'^(_Z\w+):\s*;[^\n]*\n(.*?)\.end_amdhsa_kernel', s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if flt not in name:
        continue
    code = body.split('.section')[0]
    g = lambda k: (re.findall(r'\.amdhsa_%s (\d+)' % k, body) or ['?'])[0]
    ins = [l for l in code.split('\n') if l.startswith('\t') and not l.startswith('\t.') and not l.startswith('\t;')]
    cnt = lambda k: sum(1 for l in ins if l.strip().startswith(k))
    print("%-70s vgpr %s acc_off %s scratch %s instrs %d mfma %d accvgpr %d ds_read %d ds_write %d glds %d s_nop %d waitcnt %d" % (
        name[-70:], g('next_free_vgpr'), g('accum_offset'), g('private_segment_fixed_size'), len(ins), cnt('v_mfma'),
        cnt('v_accvgpr'), cnt('ds_read'), cnt('ds_write'), cnt('global_load_lds'), cnt('s_nop'), cnt('s_waitcnt')))
