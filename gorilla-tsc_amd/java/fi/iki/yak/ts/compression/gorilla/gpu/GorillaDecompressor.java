package fi.iki.yak.ts.compression.gorilla.gpu;

/** Same surface as the reference's GorillaDecompressor (G/GorillaDecompressor.java:23-49): readPair() until it returns null. */
public class GorillaDecompressor extends BlockDecoder {
    public GorillaDecompressor(BitInput input) { super(input, Gtsc.FMT_V2, null); }

    public GorillaDecompressor(BitInput input, Predictor predictor) { super(input, Gtsc.FMT_V2, predictor); }

    public static int decodeZigZag32(final int n) { return (n >>> 1) ^ -(n & 1); }
}
