export TMPDIR=/tmp
timeout 1200 python bench.py > gpurun_out/r6_final_c2.json 2> gpurun_out/r6_final_c2.err
python tools/show_bench.py gpurun_out/r6_final_c2.json | grep -E '"value"|ms_per_step|excl_optimizer|dropin_iters|patched_full|express|host_issue|"cores"|"kind"|fits' 
