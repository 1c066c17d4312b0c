for cfg in "0 3" "0 4" "1 3" "2 3" "2 4" "3 3" "4 4"; do set -- $cfg; echo -n "wpb=$1 stages=$2: "; PGMI_ATT_WPB=$1 PGMI_ATT_STAGES=$2 timeout 100 python bench.py --layers 4 --steps 2 --warmup 1 --cpu-seconds 0 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['kernels']['attention'], d['ms_per_step'])"; done
