for d in 0 1 2 3 4 7; do echo "== DV_WINO_DBG=$d"; DV_WINO_DBG=$d python tools/wino_check.py benchonly 2>&1 | grep winograd | sed 's/direct.*//'; done
