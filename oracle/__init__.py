"""CPU oracle for the datatable groupby hot path -- TEST INFRASTRUCTURE ONLY.

May be imported only by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product (datatable_amd) never imports this package.
"""
