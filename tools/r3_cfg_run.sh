set -x
timeout -k 10 280 python -u tools/config_bench.py --moving --fuse=-1 tgv512_fp32 || exit 1
timeout -k 10 280 python -u tools/config_bench.py --moving --fuse=-1 fp32_1024 || exit 1
timeout -k 10 500 python -u tools/config_bench.py --moving --steps=20 capacity1600_fp16s || exit 1
timeout -k 10 500 python -u tools/config_bench.py --moving --steps=20 --fuse=-1 capacity1600_fp16s || exit 1
timeout -k 10 500 python -u tools/config_bench.py --moving --steps=20 --fuse=-1 --band=16,0,0 capacity1600_fp16s || exit 1
timeout -k 10 500 python -u tools/config_bench.py --moving --steps=20 --fuse=-1 --band=32,0,0 capacity1600_fp16s || exit 1
