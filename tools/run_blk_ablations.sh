for t in ${TAGS:-PLAIN NOW NOOUT NOSTORE}; do
  LAD_STAMP_LIB=$PWD/tools/libexp_blk_$t.so python tools/stamp_block.py 2>&1 | grep -v amdgpu.ids
done
