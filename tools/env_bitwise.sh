# tools/env_bitwise.sh VAR a b — the sampler's outputs under VAR=a and VAR=b must be bit-identical
R=/root/repo; O=$R/gpurun_out/env_bitwise; mkdir -p $O
env $1=$2 python $R/tools/env_bitwise.py > $O/$1_$2.txt 2>&1
env $1=$3 python $R/tools/env_bitwise.py > $O/$1_$3.txt 2>&1
if diff $O/$1_$2.txt $O/$1_$3.txt > /dev/null; then echo "$1: $2 vs $3 bit-identical"; cat $O/$1_$3.txt; else echo "$1: $2 vs $3 DIFFER"; diff $O/$1_$2.txt $O/$1_$3.txt; fi
