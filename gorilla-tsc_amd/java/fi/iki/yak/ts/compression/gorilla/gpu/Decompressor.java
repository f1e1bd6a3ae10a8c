package fi.iki.yak.ts.compression.gorilla.gpu;

/** The 1.x decompressor (G/Decompressor.java:22-42). */
public class Decompressor extends BlockDecoder {
    public Decompressor(BitInput input) { super(input, Gtsc.FMT_V1, null); }
}
