#!/bin/bash
# run every tools/attnq64_x_* variant on the judged shapes (16-bit output rows as in the product path), print time + stamps + any mismatch
for v in $(ls tools | grep attnq64_x_ | sed s/attnq64_//); do echo "=== $v"; for c in ${CASES:-8 9 12}; do Q64_OLP=1 timeout 120 tools/attnq64_$v $c 2>&1 | awk '/^----/{print} /^q64 ksplit=1 |^q64 half plan/{print; getline; print; getline; print; getline; if ($0 ~ /stamps/) print}' ; done; done
