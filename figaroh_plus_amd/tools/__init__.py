"""Mirror of ``src/figaroh/tools`` for the regressor / QR hot path (HIP-backed)."""
