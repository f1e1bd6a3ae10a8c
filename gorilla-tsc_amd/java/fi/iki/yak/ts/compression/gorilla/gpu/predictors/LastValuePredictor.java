package fi.iki.yak.ts.compression.gorilla.gpu.predictors;

import fi.iki.yak.ts.compression.gorilla.gpu.Gtsc;
import fi.iki.yak.ts.compression.gorilla.gpu.Predictor;

/** The previous value is the prediction: classic Gorilla (G/predictors/LastValuePredictor.java:10-22). */
public class LastValuePredictor implements Predictor {
    @Override
    public int kind() { return Gtsc.PRED_LAST; }
}
