mkdir -p gpurun_out
for nw in 4 8; do for v in "MI355GP_DBG_NOSYNC=1" "MI355GP_DBG_1WG=1" "MI355GP_DBG_1WG=1 MI355GP_DBG_LD0=1" "MI355GP_DBG_NOSYNC=1 MI355GP_DBG_1WG=1"; do echo "== NW=$nw $v"; env MI355GP_GEMM_NW=$nw $v python tests/gemm_probe.py 4096x4096x4096; done; done
