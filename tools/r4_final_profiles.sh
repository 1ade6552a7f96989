set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4final
mkdir -p $O
cd $R
python bench.py --steps 20 --warmup 5 --prof-dump $O/conv_launches.csv > $O/bench_line.json 2> $O/bench.err
echo bench done
python bench.py --steps 10 --warmup 3 --precision bf16 --size 512 --batch 8 --no-cpu-baseline > $O/bench_c3.json 2>/dev/null
python bench.py --steps 10 --warmup 3 --size 512 --batch 8 --no-cpu-baseline > $O/bench_512.json 2>/dev/null
python bench.py --steps 10 --warmup 3 --size 1024 --batch 4 --checkpoint --no-cpu-baseline > $O/bench_c5.json 2>/dev/null
python bench.py --steps 30 --warmup 5 --batch 1 --no-cpu-baseline > $O/bench_b1.json 2>/dev/null
python bench.py --steps 30 --warmup 5 --batch 1 --graph --no-cpu-baseline > $O/bench_b1g.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --size 512 --batch 2 --no-cpu-baseline > $O/bench_b2_512.json 2>/dev/null
echo others done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o ks -- python3 $R/bench.py --steps 30 --warmup 4 --no-cpu-baseline --no-roofline > $O/prof.log 2>&1
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
python3 $R/tools/kstats.py $O/kernel_stats.csv 34 12
