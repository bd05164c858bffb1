"""Reference path `patch_samplers` -> deephisto_amd.patch_samplers (see deephisto_amd/aliases.py)."""
