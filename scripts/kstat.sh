#!/bin/bash
# usage: scripts/kstat.sh <file.hip> [extra -D flags]: compiles one TU for gfx950 into /tmp/k16 and prints, per kernel,
# registers / spills / scratch and instruction counts that matter for the MFMA chains
set -e
SRC=$1; shift
OUT=/tmp/k16
mkdir -p $OUT
B=$(basename $SRC .hip)
cd $(dirname $SRC)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -ffp-contract=off -Wall -Wno-unused-function -Wno-unused-variable "$@" \
   -S --cuda-device-only -o $OUT/$B.s $(basename $SRC) 2>&1 | grep -v "hip-link" || true
python3 - $OUT/$B.s <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
# split per kernel
for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)s_endpgm', txt, re.S | re.M):
    name, body = m.group(1), m.group(2)
    meta = re.search(r'\.name:\s+' + re.escape(name) + r'\n(.*?)\.wavefront_size', txt, re.S)
    def cnt(p): return len(re.findall(p, body))
    VM0 = r'vmcnt\(0\)'
    md = txt[txt.find('.amdhsa_kernel ' + name):]
    vg = re.search(r'\.amdhsa_next_free_vgpr (\d+)', md).group(1)
    ag = re.search(r'\.amdhsa_accum_offset (\d+)', md)
    sp = re.search(r'; ScratchSize: (\d+)', txt[m.end():m.end()+4000])
    print(f"{name[:70]:70s} vgpr={vg} accoff={ag.group(1) if ag else '-'} scratch={sp.group(1) if sp else '?'} mfma={cnt(r'v_mfma')} "
          f"scr_ld={cnt(r'scratch_load')} scr_st={cnt(r'scratch_store')} dsr128={cnt(r'ds_read_b128')} vm0={cnt(VM0)} "
          f"bar={cnt(r's_barrier')} dma={cnt(r'global_load_lds')} vmov={cnt(r'v_mov_b32')} acc={cnt(r'v_accvgpr')} lines={body.count(chr(10))}")
PY
