#!/bin/bash
# ISA audit of the convolution kernels: compiles the instantiation units with -save-temps and prints, per
# kernel, the instruction census of the whole kernel and of its main loop (one iteration = CU chunks x NTAP taps), and the
# opcode histogram of everything behind the main loop (the epilogue).   bash tools/diag/isa_audit.sh > profiles/rNN_isa_audit.txt
set -e
root=$(cd "$(dirname "$0")/../.." && pwd)
tmp=$(mktemp -d)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form=1 -Wno-unused-function"
cd $tmp
for tu in conv_sf_inst_a conv_sf_inst_b conv_sf_inst_d conv_sf_inst_e; do
  /opt/rocm/bin/hipcc $FLAGS -I$root/atdn_vslam_amd/csrc -c $root/atdn_vslam_amd/csrc/$tu.hip -o $tu.o -save-temps=obj 2>/dev/null
done
cd $root
audit() {  # file fragment title
  echo "== $3"
  python3 tools/diag/isa_loop_census.py $tmp/$1-hip-amdgcn-amd-amdhsa-gfx950.s "$2" | sed 's/^_ZN[^ ]* //' | awk 'NR==1 || /mfma/' | sort -t"'" -k1,1 -u | head -3
  python3 - "$tmp/$1-hip-amdgcn-amd-amdhsa-gfx950.s" "$2" <<'PY'
import re, sys
from collections import Counter
txt = open(sys.argv[1]).read()
m = re.search(r"^(\S*%s\S*):[^\n]*\n" % re.escape(sys.argv[2]), txt, re.M)
body = txt[m.end():txt.index('s_endpgm', m.end())].split('\n')
labels = {}
for i, l in enumerate(body):
    mm = re.match(r'^(\.LBB\d+_\d+):', l)
    if mm: labels[mm.group(1)] = i
end = 0
for i, l in enumerate(body):
    mm = re.search(r's_cbranch_\w+ (\.LBB\d+_\d+)', l)
    if mm and mm.group(1) in labels and labels[mm.group(1)] < i and 'v_mfma' in "\n".join(body[labels[mm.group(1)]:i]): end = max(end, i)
c = Counter()
for l in body[end:]:
    l = l.strip()
    if not l or l.startswith(';') or l.startswith('.'): continue
    c[l.split()[0]] += 1
tot = sum(v for k, v in c.items() if k.startswith('v_') and not k.startswith('v_mfma'))
print("   behind the main loop (epilogue): %d vector-ALU, %d LDS, %d vector-memory, %d s_waitcnt; top opcodes: %s" % (
    tot, sum(v for k, v in c.items() if k.startswith('ds_')), sum(v for k, v in c.items() if k.startswith(('global_', 'buffer_'))),
    c.get('s_waitcnt', 0), ", ".join("%s %d" % kv for kv in c.most_common(14))))
PY
}
audit conv_sf_inst_d "conv_sf6_kernelILi8ELi16ELi128ELi1ELi4ELi1ELi5ENS_7SfGruZRELb0" "z|r ConvGRU gate, 1x5, 128-wide block (4 waves, TM = 4): one loop iteration = 2 chunks x 5 taps"
audit conv_sf_inst_d "conv_sf6_kernelILi8ELi16ELi128ELi1ELi4ELi5ELi1ENS_7SfGruZRELb0" "z|r ConvGRU gate, 5x1"
audit conv_sf_inst_d "conv_sf6_kernelILi8ELi16ELi128ELi1ELi4ELi1ELi5ENS_6SfGruQELb0" "q ConvGRU gate, 1x5"
audit conv_sf_inst_d "conv_sf6_kernelILi8ELi16ELi128ELi1ELi4ELi5ELi1ENS_6SfGruQELb0" "q ConvGRU gate, 5x1"
audit conv_sf_inst_e "conv_sf6_kernelILi8ELi16ELi128ELi1ELi4ELi3ELi3ENS_17SfFlowHeadPartialELb0" "flow head conv1 + conv2 partial sums, 3x3, 128-wide block (4 waves, two blocks per pixel tile since round 5): one loop iteration = 1 chunk x 9 taps"
audit conv_sf_inst_a "conv_sf6_kernelILi12ELi16ELi64ELi2ELi2ELi3ELi3ENS_6SfBiasILi1EEELb0ELb0" "thin 3x3, 64-wide block, 12x16 tile, plain sf store (cnet layer1, convc2, convf2)"
audit conv_sf_inst_a "conv_sf6_kernelILi8ELi16ELi128ELi1ELi4ELi3ELi3ENS_6SfBiasILi1EEELb0ELb0" "3x3, 128-wide block (motion encoder 256 -> 126)"
audit conv_sf_inst_b "conv_sf6_kernelILi12ELi16ELi64ELi2ELi2ELi3ELi3ENS_12EpiBiasStatsELb0ELb0" "fnet statistics conv, 64 channels, plain loader"
audit conv_sf_inst_b "conv_sf6_kernelILi12ELi16ELi64ELi2ELi2ELi3ELi3ENS_12EpiBiasStatsELb0ELb1" "fnet statistics conv, 64 channels, normalise-on-load"
audit conv_sf_inst_b "conv_sf6_kernelILi12ELi16ELi64ELi2ELi2ELi3ELi3ENS_17SfBiasReluAddReluELb0ELb0" "cnet residual tail (relu(res + relu(conv))), 64 channels"
rm -rf $tmp
