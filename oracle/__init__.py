"""CPU restatement of the hot path -- TEST INFRASTRUCTURE ONLY (see tb_oracle.c header).

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; the
product package tennisbot_rl_amd never imports this.
"""
from .cpu_oracle import OracleBatch, build, lib_path, philox4x32, query_box, query_goal, query_racket, query_racket_ground  # noqa: F401
