#!/bin/bash
# tools/quick.sh <layout> [bench args] -> one compact line (kernel ms, frac, extras)
L=$1; shift
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --layout $L "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$L', '$VR_CORE_LIB', d['config']['kernel'], 'kernel_ms', d['kernel_ms'], 'frac', d['roofline']['frac'], {k:(v['kernel_ms'] if isinstance(v,dict) else v) for k,v in d.get('extras',{}).items()})"
