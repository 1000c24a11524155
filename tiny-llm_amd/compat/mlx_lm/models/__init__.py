"""`mlx_lm.models` names the reference's tests use as oracles (facade over torch; see ../../README.md)."""
