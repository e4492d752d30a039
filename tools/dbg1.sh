cd $GRAFT_REPO_ROOT
for L in a2 a2q; do
echo "== fused + stream tests lib_$L"
IQD_LIB=$PWD/tmp_variants/lib_$L.so python -m pytest tests/test_gpu_scale.py tests/test_gpu_stream.py tests/test_gpu_wbfm.py -x -q 2>&1 | grep -v "^  File\|^Extension" | tail -4
done
tools/abn.sh 5 "" tmp_variants/lib_norunptr.so tmp_variants/lib_a4.so tmp_variants/lib_a2.so tmp_variants/lib_a2q.so
