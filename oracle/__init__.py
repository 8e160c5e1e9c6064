"""oracle/ -- CPU restatements used ONLY as checkers.

TEST INFRASTRUCTURE.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this package; nothing under
``pika_amd/`` does (tests/test_no_oracle_in_product.py enforces it).
"""
