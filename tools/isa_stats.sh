#!/bin/bash
# instruction mix of one kernel: tools/isa_stats.sh <mangled-name-prefix>   (no GPU needed)
K=${1:-_Z16turn_frac_kernelILb0EEv7DevViewi}
D=/tmp/pedn_isa; mkdir -p $D
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math --cuda-device-only -S -o $D/pedn.s $(dirname $0)/../pednstream_amd/csrc/pedn_hip.hip 2>/dev/null || { echo compile failed; exit 1; }
awk "/^$K:/,/s_endpgm/" $D/pedn.s > $D/k.s
echo "$K: $(grep -c -E '^\s+[vsdgb]_|^\s+(global|flat|scratch|buffer|ds)_' $D/k.s) instructions"
for k in v_div_scale_f64 v_rcp_f64 v_fma_f64 v_mul_f64 v_add_f64 v_cndmask v_readlane v_writelane s_cbranch s_waitcnt global_load global_store ds_read ds_write scratch_; do echo "  $k $(grep -c $k $D/k.s)"; done
