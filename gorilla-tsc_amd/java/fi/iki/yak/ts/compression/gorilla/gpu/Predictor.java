package fi.iki.yak.ts.compression.gorilla.gpu;

/**
 * The value predictor of a 2.x compressor / decompressor (G/Predictor.java:22-37).  Prediction
 * happens on the GPU; an implementation only says which predictor the library is to use
 * (gtsc_set_predictor).
 */
public interface Predictor {
    /** Gtsc.PRED_LAST or Gtsc.PRED_DFCM. */
    int kind();

    /** Table size of DifferentialFCM; ignored otherwise. */
    default int tableSize() { return 0; }
}
