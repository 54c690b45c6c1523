cd /root/repo
mkdir -p gpurun_out/r04h
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r04h/tests.log
for v in 0; do echo "u16 variant $v: $(timeout 300 python tools/tri_ms.py $v orbit6 1024 2 2>&1 | tail -1)"; done > gpurun_out/r04h/orbit.txt 2>&1
for v in 0; do echo "u16 variant $v both: $(timeout 300 python tools/tri_ms.py $v both 1024 2 2>&1 | tail -1)"; done >> gpurun_out/r04h/orbit.txt 2>&1
for v in 0; do echo "u8 variant $v: $(timeout 300 python tools/tri_ms.py $v orbit6 1024 1 2>&1 | tail -1)"; done >> gpurun_out/r04h/orbit.txt 2>&1
cat gpurun_out/r04h/tests.log gpurun_out/r04h/orbit.txt
