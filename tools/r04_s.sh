# a short exact division on top of pm_rcp (tests/micro/div_check.hip): differences against a / b, latency
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -w -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fgpu-flush-denormals-to-zero tests/micro/div_check.hip -o /tmp/div_check || exit 1
timeout -k 10 300 /tmp/div_check > $O/s_div.log 2>&1; cat $O/s_div.log
