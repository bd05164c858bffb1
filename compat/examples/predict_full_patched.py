"""Reference path examples/predict_full_patched.py."""
from deephisto_amd.examples.predict_full_patched import (ImagePredictorPatched, batch_predictor, load_model, main,  # noqa: F401
                                                         perform_and_save_visualizations, predict_full_patched)

if __name__ == "__main__":
    main()
