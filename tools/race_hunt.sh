#!/bin/bash
mkdir -p gpurun_out/r4f
T="tests/test_inference_gpu.py::test_inference_batched_equals_single_image_and_host_path"
for rep in 1 2; do
  echo "== default layout rep $rep" >> gpurun_out/r4f/race.txt; timeout -k 10 200 python -m pytest $T -x -q 2>&1 | tail -2 >> gpurun_out/r4f/race.txt
  echo "== ISEGMI_STREAM_PLACEMENT=0 rep $rep" >> gpurun_out/r4f/race.txt; ISEGMI_STREAM_PLACEMENT=0 timeout -k 10 200 python -m pytest $T -x -q 2>&1 | tail -2 >> gpurun_out/r4f/race.txt
  echo "== side0 on main's queue rep $rep" >> gpurun_out/r4f/race.txt; ISEGMI_STREAM_LAYOUT=0021123012 timeout -k 10 200 python -m pytest $T -x -q 2>&1 | tail -2 >> gpurun_out/r4f/race.txt
  echo "== copy alone (D) rep $rep" >> gpurun_out/r4f/race.txt; ISEGMI_STREAM_LAYOUT=0321123013 timeout -k 10 200 python -m pytest $T -x -q 2>&1 | tail -2 >> gpurun_out/r4f/race.txt
  echo "== tail alone: side2/hs2 elsewhere rep $rep" >> gpurun_out/r4f/race.txt; ISEGMI_STREAM_LAYOUT=0323123032 timeout -k 10 200 python -m pytest $T -x -q 2>&1 | tail -2 >> gpurun_out/r4f/race.txt
done
true
