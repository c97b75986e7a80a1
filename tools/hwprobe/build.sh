#!/bin/bash
# Builds the hardware probes of this directory (binaries and the generated pv_*.h instruction streams are not tracked):
#   tools/hwprobe/build.sh [potf2_probe|diag128_probe|valu_probe|vmcnt_order ...]     (run from the repo root; default: all)
# pv_*.h are variants of the 16 x 16 potf2 stream from tools/gen_potf2.py (the product's csrc/potf2_asm.h is its default output):
#   fused    : the shipped stream (v_fmac_f64_dpp, plain v_rsq_f64)        pad      : + one wait state in front of every VALU
#   prsq     : v_rsq_f64_dpp (assembles; gfx950 returns garbage for it)    pfmac    : v_mov_b64_dpp + v_fmac_f64, v_rsq_f64_dpp
#   plain    : v_mov_b64_dpp + v_fmac_f64                                  plainpad : plain + one wait state per VALU
set -e
H=tools/hwprobe
gen() { python tools/gen_potf2.py --name potf2_v_$1 ${@:2} > $H/pv_$1.h; }
gen fused
gen pad --pad 1
gen prsq --dpp-rsq
gen pfmac --plain-fmac --dpp-rsq
gen plain --plain-fmac
gen plainpad --plain-fmac --pad 1
for p in ${@:-potf2_probe diag128_probe valu_probe vmcnt_order}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -I gpy_amd/csrc -I $H $H/$p.hip -o $H/$p
  echo "built $H/$p"
done
