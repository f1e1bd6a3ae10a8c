package fi.iki.yak.ts.compression.gorilla.gpu.predictors;

import fi.iki.yak.ts.compression.gorilla.gpu.Gtsc;
import fi.iki.yak.ts.compression.gorilla.gpu.Predictor;

/** Differential finite context method with a table of `size` entries (G/predictors/DifferentialFCM.java:23-46). */
public class DifferentialFCM implements Predictor {
    private final int size;

    public DifferentialFCM(int size) {
        if (size <= 0) throw new IllegalArgumentException("Size cannot be negative and should be power of 2");
        this.size = size;   // the library rounds it up to a power of two, as the reference's constructor does
    }

    @Override
    public int kind() { return Gtsc.PRED_DFCM; }

    @Override
    public int tableSize() { return size; }
}
