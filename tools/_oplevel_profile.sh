# Evidence for the operator-level (fallback) paths and the shape map: everything lands in gpurun_out/oplevel/
mkdir -p gpurun_out/oplevel
python tools/shape_cliff_timing.py 2>&1 | grep "N=" > gpurun_out/oplevel/shape_sweep_f64.txt
python tools/shape_cliff_timing.py --dtype f32 --cases IQU:35,IQUV:27,IQUV:43,IQU:61,I:127,IQUV:47,IQUV:51 2>&1 | grep "N=" > gpurun_out/oplevel/shape_sweep_f32.txt
VSM_NO_GEMM_LDS=1 python tools/shape_cliff_timing.py --cases IQU:41,IQUV:43 2>&1 | grep "N=" > gpurun_out/oplevel/shape_sweep_f64_old_gemm.txt
VSM_NO_GEMM_LDS=1 python tools/shape_cliff_timing.py --dtype f32 --cases IQUV:27,IQUV:43,IQUV:51 2>&1 | grep "N=" > gpurun_out/oplevel/shape_sweep_f32_old_gemm.txt
python tools/gemm_timing.py 2>&1 | grep "N=" > gpurun_out/oplevel/gemm_f64.txt
python tools/gemm_timing.py --dtype f32 2>&1 | grep "N=" > gpurun_out/oplevel/gemm_f32.txt
python tools/inv_timing.py 2>&1 | grep "N=" > gpurun_out/oplevel/inv_f64.txt
python tools/inv_timing.py --dtype f32 2>&1 | grep "N=" > gpurun_out/oplevel/inv_f32.txt
python tools/thermal_timing.py 2>&1 | grep "N=" > gpurun_out/oplevel/thermal.txt
bash tools/_cliffstats.sh f64n96 --cases IQUV:43 --no-lin > gpurun_out/oplevel/kernel_stats_f64_n96_forward.txt 2>&1
bash tools/_cliffstats.sh f32n96lin --dtype f32 --cases IQUV:43 > gpurun_out/oplevel/kernel_stats_f32_n96_lin.txt 2>&1
bash tools/_rw39.sh > gpurun_out/oplevel/kernel_stats_raman_n39.txt 2>&1
cat gpurun_out/oplevel/*.txt
