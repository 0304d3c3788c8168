# usage: ab_env.sh VAR VALUE -> bench with VAR=VALUE and with the default, twice each, same box
run() { python bench.py --no-cpu-baseline | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'])"; }
for i in 1 2; do echo -n "$1=$2: "; env $1=$2 python bench.py --no-cpu-baseline | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'])"; echo -n "default: "; run; done
