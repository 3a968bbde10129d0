"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the nas_3d_unet hot path.

Nothing under ``oracle/`` is product code.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker / the reported CPU baseline.  The product
package ``nas_3d_unet_amd`` never imports this package and has no CPU
fallback: it fails loudly when the HIP library is missing.

Parity status: PINNED.  The reference ships no tests or golden vectors
(SURVEY.md section 4), so the oracle is pinned against outputs of the
reference itself: ``tests/golden/make_golden.py`` imports the six reference
model files from /root/reference in the build container and writes small
fixtures (``tests/golden/*.npz``); ``tests/test_oracle_golden.py`` checks this
restatement against every one of them.
"""
from .ref_path import *  # noqa: F401,F403
