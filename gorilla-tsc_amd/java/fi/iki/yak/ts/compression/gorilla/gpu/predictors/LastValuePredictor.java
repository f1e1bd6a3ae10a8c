package fi.iki.yak.ts.compression.gorilla.gpu.predictors;

import fi.iki.yak.ts.compression.gorilla.gpu.Predictor;

/** The previous value is the prediction: classic Gorilla (G/predictors/LastValuePredictor.java:10-22). */
public class LastValuePredictor implements Predictor {
    private long storedVal = 0;

    @Override
    public void update(long value) { this.storedVal = value; }

    @Override
    public long predict() { return storedVal; }
}
