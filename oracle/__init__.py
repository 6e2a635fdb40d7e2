"""CPU oracle for the libgpuhash hot path -- TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg; the product package (megakv_amd) must never import it.
"""
