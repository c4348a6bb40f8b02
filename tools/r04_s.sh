# exhaustive check of a short correctly rounded reciprocal against the compiler's 1.0f / x (tests/micro/rcp_exhaustive.hip)
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -w -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fgpu-flush-denormals-to-zero tests/micro/rcp_exhaustive.hip -o /tmp/rcp_ex
timeout -k 10 200 /tmp/rcp_ex > $O/s_rcp.log 2>&1; cat $O/s_rcp.log
