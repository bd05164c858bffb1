"""Reference path examples/sample_full_dense.py."""
from deephisto_amd.examples.sample_full_dense import main  # noqa: F401

if __name__ == "__main__":
    main()
