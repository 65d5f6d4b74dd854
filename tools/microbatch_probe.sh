set -e
mkdir -p gpurun_out/mb
for b in 8 4 2 1; do python tools/conv_report.py $b 0 maskrcnn fp16 101 > gpurun_out/mb/r101f16_bs$b.txt; done
for b in 8 4 2; do python tools/conv_report.py $b 0 yolact > gpurun_out/mb/yolact_bs$b.txt; done
for b in 2 1; do python tools/conv_report.py $b 0 maskrcnn > gpurun_out/mb/maskrcnn_bs$b.txt; done
for f in gpurun_out/mb/*.txt; do echo $f; python tools/stage_table.py $f | grep -E "stem|res2|res3|res4|total"; done
