"""CPU oracle for the DDNM hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is product code.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker / reported baseline.  The product path
(``ddnm_amd``) never imports this package and fails loudly when its HIP
library is missing.

Contents
--------
* ``ref_import``   -- imports the real reference from ``/root/reference`` (only
                      where that tree exists, i.e. the build container) with the
                      three shims SURVEY.md section 8(c) lists.  Used to pin the
                      restatement and to generate ``tests/golden``.
* ``weights``      -- deterministic seeded state-dicts (reference key names).
* ``schedule``     -- restated beta / alpha-bar / jump schedule.
* ``operators``    -- restated degradation operators (direct math form).
* ``unet_celeba``  -- restated ``guided_diffusion/models.py::Model`` forward.
* ``unet_adm``     -- restated ``guided_diffusion/unet.py::UNetModel`` forward.
* ``sampler``      -- restated ``ddnm_diffusion`` / simplified loop.

Parity pinning: the reference ships NO tests, golden vectors or KATs for this
path.  The restatement is therefore pinned against outputs
of the reference itself, generated in the build container by
``tests/golden/make_golden.py`` (imports ``/root/reference``) and committed
under ``tests/golden/*.npz``.
"""
