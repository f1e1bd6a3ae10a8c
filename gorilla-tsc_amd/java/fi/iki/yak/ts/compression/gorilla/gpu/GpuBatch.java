package fi.iki.yak.ts.compression.gorilla.gpu;

import java.lang.foreign.Arena;
import java.lang.foreign.MemorySegment;
import java.util.ArrayList;
import java.util.List;

import static java.lang.foreign.ValueLayout.JAVA_INT;
import static java.lang.foreign.ValueLayout.JAVA_LONG;

/**
 * Collects closed blocks and codes them with one library call per format and predictor.
 * A foreign call per point would cost more than the codec, so addValue() only buffers; flush() does the work.
 * A batch made with several contexts (normally one per GPU of the node, Gtsc.createContext(device)) spreads its
 * blocks over them through gtsc_encode_host_multi: contiguous runs of blocks with about equal numbers of points,
 * each through its context's pipeline on a host thread of its own.
 */
public final class GpuBatch {
    private final List<BlockEncoder> closed = new ArrayList<>();
    private final MemorySegment[] contexts;   // null: the calling thread's own context

    public GpuBatch() { this.contexts = null; }

    /** A batch over the given contexts (Gtsc.createContext), e.g. one per GPU. */
    public GpuBatch(MemorySegment... contexts) { this.contexts = contexts.clone(); }

    void add(BlockEncoder c) { closed.add(c); }

    /** Codes every closed block; each block's output then holds its image. */
    public void flush() {
        while (!closed.isEmpty()) {
            BlockEncoder k = closed.get(0);
            List<BlockEncoder> same = new ArrayList<>(), rest = new ArrayList<>();
            for (BlockEncoder b : closed)
                (b.format == k.format && b.predictor == k.predictor && b.tableSize == k.tableSize ? same : rest).add(b);
            closed.clear();
            closed.addAll(rest);
            code(same, true);
        }
    }

    /** One library call for blocks of one format and predictor: closed blocks, or snapshots of open ones. */
    void code(List<BlockEncoder> blocks, boolean close) {
        int S = blocks.size();
        long total = 0, cap = 0;
        for (BlockEncoder c : blocks) { total += c.size; cap += Gtsc.maxWords(c.size); }
        BlockEncoder first = blocks.get(0);
        try (Arena a = Arena.ofConfined()) {
            MemorySegment blockTs = a.allocate(JAVA_LONG, S);
            MemorySegment ptOff = a.allocate(JAVA_LONG, S + 1L);
            MemorySegment ts = a.allocate(JAVA_LONG, Math.max(total, 1));
            MemorySegment val = a.allocate(JAVA_LONG, Math.max(total, 1));
            MemorySegment words = a.allocate(JAVA_LONG, cap);
            MemorySegment outOff = a.allocate(JAVA_LONG, S + 1L);
            MemorySegment nbits = a.allocate(JAVA_LONG, S);
            long p = 0;
            for (int s = 0; s < S; s++) {
                BlockEncoder c = blocks.get(s);
                blockTs.setAtIndex(JAVA_LONG, s, c.blockTimestamp);
                ptOff.setAtIndex(JAVA_LONG, s, p);
                MemorySegment.copy(c.ts, 0, ts, JAVA_LONG, p * 8, c.size);
                MemorySegment.copy(c.val, 0, val, JAVA_LONG, p * 8, c.size);
                p += c.size;
            }
            ptOff.setAtIndex(JAVA_LONG, S, p);
            MemorySegment[] ctxs = contexts != null ? contexts : new MemorySegment[] {Gtsc.CONTEXT.get()};
            MemorySegment handles = a.allocate(java.lang.foreign.ValueLayout.ADDRESS, ctxs.length);
            int rc = Gtsc.OK;
            for (int k = 0; k < ctxs.length; k++) {
                handles.setAtIndex(java.lang.foreign.ValueLayout.ADDRESS, k, ctxs[k]);
                rc = (int) Gtsc.SET_PREDICTOR.invokeExact(ctxs[k], first.predictor, first.tableSize);
                if (rc != Gtsc.OK) throw new IllegalStateException(Gtsc.lastError());
            }
            try {
                rc = (int) Gtsc.ENCODE_HOST_MULTI.invokeExact(ctxs.length, handles, first.format, close ? 0 : Gtsc.ENC_NO_CLOSE, (long) S,
                        blockTs, ts, val, ptOff, 0L, words, cap, outOff, nbits);
            } finally {
                for (MemorySegment c : ctxs) { int ignored = (int) Gtsc.SET_PREDICTOR.invokeExact(c, Gtsc.PRED_LAST, 0); }
            }
            if (rc != Gtsc.OK) throw new IllegalStateException(Gtsc.lastError());
            for (int s = 0; s < S; s++) {
                long b = outOff.getAtIndex(JAVA_LONG, s), e = outOff.getAtIndex(JAVA_LONG, s + 1L);
                long[] image = new long[(int) (e - b)];
                MemorySegment.copy(words, JAVA_LONG, b * 8, image, 0, image.length);
                BlockEncoder c = blocks.get(s);
                c.deliver(image, nbits.getAtIndex(JAVA_LONG, s), close);
                if (close) c.coded = true;
            }
        } catch (RuntimeException e) {
            throw e;
        } catch (Throwable t) {
            throw new IllegalStateException(t);
        }
    }

    /**
     * Decodes one image; returns {timestamps, raw values} (the arrays' length is the pair count), with a third,
     * empty array appended when the image ended inside a value (the pairs before that point are still returned).
     */
    static long[][] decode(int format, int predictor, int tableSize, long[] image) {
        long cap = image.length * 32L + 1;  // a pair takes at least two bits
        try (Arena a = Arena.ofConfined()) {
            MemorySegment words = a.allocate(JAVA_LONG, Math.max(image.length, 1));
            MemorySegment.copy(image, 0, words, JAVA_LONG, 0, image.length);
            MemorySegment inOff = a.allocate(JAVA_LONG, 2);
            inOff.setAtIndex(JAVA_LONG, 1, image.length);
            MemorySegment ts = a.allocate(JAVA_LONG, cap), val = a.allocate(JAVA_LONG, cap);
            MemorySegment n = a.allocate(JAVA_LONG), st = a.allocate(JAVA_INT);
            MemorySegment ctx = Gtsc.CONTEXT.get();
            int rc = (int) Gtsc.SET_PREDICTOR.invokeExact(ctx, predictor, tableSize);
            if (rc != Gtsc.OK) throw new IllegalStateException(Gtsc.lastError());
            try {
                rc = (int) Gtsc.DECODE_HOST.invokeExact(ctx, format, 1L, words, inOff, ts, val, MemorySegment.NULL, cap, n, st);
            } finally {
                int ignored = (int) Gtsc.SET_PREDICTOR.invokeExact(ctx, Gtsc.PRED_LAST, 0);
            }
            if (rc != Gtsc.OK && rc != Gtsc.ERR_STREAM) throw new IllegalStateException(Gtsc.lastError());
            int status = st.get(JAVA_INT, 0);
            boolean damaged = (status & (Gtsc.ST_TRUNCATED | Gtsc.ST_MALFORMED)) != 0;
            int count = (int) n.get(JAVA_LONG, 0);
            long[][] r = new long[damaged ? 3 : 2][];
            r[0] = new long[count];
            r[1] = new long[count];
            if (damaged) r[2] = new long[0];
            MemorySegment.copy(ts, JAVA_LONG, 0, r[0], 0, count);
            MemorySegment.copy(val, JAVA_LONG, 0, r[1], 0, count);
            return r;
        } catch (RuntimeException e) {
            throw e;
        } catch (Throwable t) {
            throw new IllegalStateException(t);
        }
    }
}
