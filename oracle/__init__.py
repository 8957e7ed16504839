"""CPU oracle for the OpenProvence forward path -- TEST INFRASTRUCTURE, never imported by the product.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import this
package, and only as the checker / reported baseline.
"""
