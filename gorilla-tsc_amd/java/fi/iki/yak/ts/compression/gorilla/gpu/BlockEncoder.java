package fi.iki.yak.ts.compression.gorilla.gpu;

import java.util.Arrays;
import java.util.List;

/** What GorillaCompressor (2.x) and Compressor (1.x) share: points are buffered, the GPU codes the block. */
abstract class BlockEncoder {
    final long blockTimestamp;
    final BitOutput out;
    final int format, predictor, tableSize;
    private final GpuBatch batch;
    private long written;   // bits of the stream the output has received
    long[] ts = new long[128], val = new long[128];
    int size;
    private boolean closed;
    boolean coded;

    BlockEncoder(long timestamp, BitOutput output, GpuBatch batch, int format, Predictor p) {
        this.blockTimestamp = timestamp;
        this.out = output;
        this.batch = batch;
        this.format = format;
        // which of the GPU's predictors the object stands for
        if (p == null || p instanceof fi.iki.yak.ts.compression.gorilla.gpu.predictors.LastValuePredictor) {
            this.predictor = Gtsc.PRED_LAST;
            this.tableSize = 0;
        } else if (p instanceof fi.iki.yak.ts.compression.gorilla.gpu.predictors.DifferentialFCM) {
            this.predictor = Gtsc.PRED_DFCM;
            this.tableSize = ((fi.iki.yak.ts.compression.gorilla.gpu.predictors.DifferentialFCM) p).tableSize();
        } else {
            throw new IllegalArgumentException("only LastValuePredictor and DifferentialFCM run on the GPU");
        }
        if (output instanceof TrackedOutput) ((TrackedOutput) output).writer = this;
    }

    /**
     * Writes stream bits [written, nbits) of the image into the output through the reference's interface, whole words
     * where it can; {@code last}: the block is closed, flush() follows as in the reference's close().
     */
    void deliver(long[] image, long nbits, boolean last) {
        while (written < nbits) {
            long w = image[(int) (written >>> 6)];
            int at = (int) (written & 63), room = 64 - at;
            int n = (int) Math.min(nbits - written, room);
            out.writeBits(n == 64 ? w : (w >>> (room - n)) & ((1L << n) - 1), n);
            written += n;
        }
        if (last) out.flush();
    }

    public void addValue(long timestamp, long value) {
        if (closed) throw new IllegalStateException("addValue after close");
        if (size == ts.length) {
            ts = Arrays.copyOf(ts, size * 2);
            val = Arrays.copyOf(val, size * 2);
        }
        ts[size] = timestamp;
        val[size++] = value;
    }

    public void addValue(long timestamp, double value) { addValue(timestamp, Double.doubleToRawLongBits(value)); }

    /** Ends the block.  Without a batch it is coded here; with one, on GpuBatch.flush(). */
    public void close() {
        if (closed) return;
        closed = true;
        GpuBatch b = batch != null ? batch : new GpuBatch();
        b.add(this);
        if (batch == null) b.flush();
    }

    /** getLongArray() / getByteBuffer() before the block was coded: a snapshot of the open block, or the batch's flush. */
    void bringOutputUpToDate() {
        if (coded) return;
        if (closed) {
            if (batch != null) batch.flush();
            return;
        }
        new GpuBatch().code(List.of(this), false);
    }
}
