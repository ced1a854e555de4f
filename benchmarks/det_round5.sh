timeout 900 python -m pytest tests/test_node_ops_gpu.py tests/test_node_fused_gpu.py tests/test_revgcn.py -q -m gpu -p no:cacheprovider 2>&1 | tail -3
python tests/guard_alloc/revgcn_sequence.py --winner 1 --rows modelfile_fused,modelfile_fused_graph --steps 6 --replays 10 2>&1 | grep "ms per"
