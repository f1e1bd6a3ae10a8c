package fi.iki.yak.ts.compression.gorilla.gpu;

/** The 1.x compressor (G/Compressor.java:24-84): same surface, format GTSC_FMT_V1. */
public class Compressor extends BlockEncoder {
    public Compressor(long timestamp, BitOutput output) { super(timestamp, output, null, Gtsc.FMT_V1, null); }

    public Compressor(long timestamp, BitOutput output, GpuBatch batch) { super(timestamp, output, batch, Gtsc.FMT_V1, null); }
}
