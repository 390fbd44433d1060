"""Mirror of ``src/figaroh/identification`` for the LS / WLS / sigma part of the hot path (HIP-backed)."""
