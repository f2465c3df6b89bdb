#!/usr/bin/env bash
# tools/shape_scan.sh <variant> <steps> <shape index> ...: bench.py's figure for each of the named tile shapes (tools/tune.py lists
# them) in turn, three rounds on one box (GPU box; tuning aid: the timed steps see the drivers' own fields, not tune.py's bounded ones)
V="$1"; S="$2"; shift 2
for rep in 1 2 3; do for sh in "$@"; do echo -n "$V steps=$S shape $sh: "; timeout -k 10 200 python3 bench.py --variant "$V" --steps "$S" --shape $sh --no-cpu --no-standing-start 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["roofline"]["launch_us"], d["config"]["kernel"][:90])'; done; done
