"""CPU oracle package -- TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; the product package
`uit_mobile_amd` never imports it."""
