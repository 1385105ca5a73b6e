"""Same import path and names as the reference's `baselines/fastkv/utils.py` (FastKVCluster, compress_fastkv,
init_fastkv, repeat_kv -- /root/reference/baselines/fastkv/utils.py:13-138); the implementation is the MI355X one."""
from fastkv_amd.cluster import FastKVCluster, compress_fastkv, init_fastkv, repeat_kv  # noqa: F401
