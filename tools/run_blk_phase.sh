python tools/bench_block.py --rounds 3 --only fused
LAD_HIP_LIB=$PWD/tools/libexp_blk_INPHASE.so python tools/bench_block.py --rounds 3 --only fused
python tools/bench_block.py --rounds 2
