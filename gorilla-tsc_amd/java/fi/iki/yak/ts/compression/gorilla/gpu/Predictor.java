package fi.iki.yak.ts.compression.gorilla.gpu;

/**
 * The value predictor of a 2.x compressor / decompressor: the reference's interface (G/Predictor.java:22-37).
 * The prediction itself happens on the GPU, which runs the reference's two predictors, LastValuePredictor and
 * DifferentialFCM; a compressor given any other implementation throws IllegalArgumentException.
 */
public interface Predictor {
    /** Give the real value. */
    void update(long value);

    /** Predicts the next value. */
    long predict();
}
