for cfg in "16 0" "4 2" "4 4" "4 8" "2 4" "2 8" "8 2" "8 4" "4 16"; do set -- $cfg; CVO_FIRST_U=$1 CVO_FIRST_CHUNKS=$2 python scripts/batch_time.py 2>&1 | tail -1; done
