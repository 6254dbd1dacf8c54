"""CPU oracle for the sampling + decode hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and only as the checker / reported CPU baseline.  The
product path (``video-llamagen_amd``) never imports from here and fails loudly
when ``libvlg.so`` is missing.
"""
