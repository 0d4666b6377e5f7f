cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/wt_f gpurun_out/wt_w
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/wt_f -- python3 tools/wgrad_traffic.py run > /dev/null 2> gpurun_out/wt_f.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/wt_w -- python3 tools/wgrad_traffic.py run > /dev/null 2> gpurun_out/wt_w.err
python3 tools/wgrad_traffic.py report gpurun_out/wt_f gpurun_out/wt_w > gpurun_out/${TAG:-r05o}_wgrad_traffic_layers.txt 2>&1
cat gpurun_out/${TAG:-r05o}_wgrad_traffic_layers.txt
rm -rf gpurun_out/wt_f gpurun_out/wt_w
