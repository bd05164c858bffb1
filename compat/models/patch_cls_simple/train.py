"""Reference entry point `python -m models.patch_cls_simple.train [--extract_test]` (train.py:304-315)."""
from deephisto_amd.models.patch_cls_simple.train import main, prepare_test_patches, save_plot, train  # noqa: F401

if __name__ == "__main__":
    main()
