"""CPU oracle of the RaCFormer decoder hot path -- TEST INFRASTRUCTURE, not product code.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` / parity leg import this
package; nothing under ``racformer_amd/`` does.  See ``restate.py`` (torch-CPU restatement, pinned by the
goldens in ``tests/golden/``) and ``gather_ref.c`` (plain-C gathers, built by ``oracle/Makefile``)."""
