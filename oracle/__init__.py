"""CPU oracle for the zeroNoteSamba pretext hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker / the timed CPU baseline.
The product path (``zeronotesamba_amd``) never imports this package and fails
loudly when the HIP library is missing.

Pieces (each cites the reference file:line it restates):

* ``cnn_ref``    -- _CNN / DS_CNN / Pretext_CNN / Down_CNN (models/models.py:7-150)
                    on stock ``torch.nn.functional`` CPU ops.  PINNED against the
                    imported reference (tests/golden/make_golden.py).
* ``ntxent_ref`` -- NT-Xent (models/loss_functions.py:24-55), closed form.  PINNED.
* ``step_ref``   -- one pretext step (pretext.py:475-492) and one fine-tune step
                    (epochs.py:48-63).  PINNED.
* ``vqt_ref``    -- float64 numpy restatement of librosa-0.8.1 ``vqt``/``cqt``
                    (processing/input_rep.py:11-57).  PARITY UNPINNED: librosa,
                    resampy and numba are not installed here and the reference
                    ships no golden vectors for this function.
* ``dropout_ref``-- the counter-based keep mask the HIP epilogues use, restated
                    in numpy so that train-mode runs can be compared bit-for-bit.
"""
