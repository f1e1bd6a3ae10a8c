package fi.iki.yak.ts.compression.gorilla.gpu;

/**
 * Same constructors and methods as fi.iki.yak.ts.compression.gorilla.GorillaCompressor
 * (G/GorillaCompressor.java:28-37,49-71,84-89), coded on the GPU.
 */
public class GorillaCompressor extends BlockEncoder {
    public static final int FIRST_DELTA_BITS = 27;

    public GorillaCompressor(long timestamp, BitOutput output) { super(timestamp, output, null, Gtsc.FMT_V2, null); }

    public GorillaCompressor(long timestamp, BitOutput output, Predictor predictor) { super(timestamp, output, null, Gtsc.FMT_V2, predictor); }

    /** Attached to a batch: close() queues the block, GpuBatch.flush() codes all queued blocks in one call. */
    public GorillaCompressor(long timestamp, BitOutput output, GpuBatch batch) { super(timestamp, output, batch, Gtsc.FMT_V2, null); }

    public GorillaCompressor(long timestamp, BitOutput output, Predictor predictor, GpuBatch batch) {
        super(timestamp, output, batch, Gtsc.FMT_V2, predictor);
    }

    public static int encodeZigZag32(final int n) { return (n << 1) ^ (n >> 31); }
}
